"""ORACLE (test infrastructure only): numpy restatement of the Kaldi post-step the reference runs behind
extract.py (egs/voxceleb/v1/nnet/run_extract_embeddings.sh:80-103):

    ivector-normalize-length --scaleup=false scp:xvector.scp ark:- | ivector-mean ark:spk2utt ark:- ark:- ark,t:num_utts.ark | ...

Kaldi is an external dependency of the reference ($KALDI_ROOT, not vendored, no pinned version) and is absent
here, so this follows the published algorithms of ivector-normalize-length.cc (ratio = ||v|| or ||v||/sqrt(dim);
a zero vector is warned about and written unchanged) and ivector-mean.cc (per speaker: Vector<BaseFloat> sum of the
utterances found, in spk2utt order, Scale(1.0 / utt_count); utterances without a vector are skipped with a
warning, speakers left with none are not written).  **parity unpinned**: no Kaldi binary, test or fixture is
available to pin it."""
import numpy as np


def normalize_length(x, scaleup=False):
    """[n, dim] float32 -> rows divided by their L2 norm (norm / sqrt(dim) with scaleup); norm in double."""
    x = np.asarray(x, dtype=np.float32)
    norm = np.sqrt(np.sum(np.square(x.astype(np.float64)), axis=1))
    ratio = norm / np.sqrt(x.shape[1]) if scaleup else norm
    inv = np.where(ratio == 0.0, 1.0, 1.0 / np.where(ratio == 0.0, 1.0, ratio))
    return (x.astype(np.float64) * inv[:, None]).astype(np.float32)


def speaker_mean(vectors, spk2utt):
    """vectors: {utt: [dim] float32}; spk2utt: [(spk, [utt, ...])] in file order.
    -> ([(spk, mean float32)], [(spk, count)]) for the speakers with at least one vector."""
    means, counts = [], []
    for spk, utts in spk2utt:
        acc, n = None, 0
        for u in utts:
            if u not in vectors:
                continue
            v = np.asarray(vectors[u], dtype=np.float32)
            acc = v.copy() if acc is None else (acc + v).astype(np.float32)      # float32 adds, spk2utt order
            n += 1
        if n == 0:
            continue
        means.append((spk, (acc * np.float32(1.0 / n)).astype(np.float32)))
        counts.append((spk, n))
    return means, counts
