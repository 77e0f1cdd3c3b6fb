"""GPU: size-independent properties at the BASELINE sizes (256 utterances x 300 frames, and a
ragged T~U[200,1000] pack), where the float64 oracle is too slow to run on everything:

  * determinism        -- two runs of the same batch are bit-identical (no atomics anywhere,
                          fused pooling merges its partials in a fixed order);
  * batch invariance   -- frame-level outputs are bit-identical wherever the utterance sits in the
                          pack (every output row depends only on its own input rows, fixed K order);
                          the pooled embedding agrees to <= 1e-6 (the fused pooling groups an
                          utterance's frames by 64-row tiles, so the fp32 summation tree -- not the
                          math -- depends on the pack position);
  * precision ladder   -- bf16x3 agrees with the exact fp32-MFMA path to <= 2e-5 on every utterance
                          (and a sample of both is checked against the oracle at <= 1e-4);
  * chunk equivalence  -- the driver's chunk/weight/average path on a long utterance equals the
                          hand-computed length-weighted mean of per-chunk embeddings.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make(precision, params=None, dim=30):
    from tf_kaldi_speaker_amd import synth
    from tf_kaldi_speaker_amd.params import Params
    from tf_kaldi_speaker_amd.trainer import Trainer
    params = dict(params or synth.TDNN_STAT_PARAMS)
    weights = synth.synth_weights(params, dim, seed=0)
    tr = Trainer(Params(**params), None, dim, single_cpu=True, device=0, precision=precision)
    tr.build("predict")
    tr.load_weights(weights)
    return tr, weights, params


def _run(tr, utts, node=None):
    import torch
    lens = [u.shape[0] for u in utts]
    packed = torch.from_numpy(np.concatenate(utts, axis=0)).cuda()
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    return tr.predict_packed(packed, offsets, node=node).cpu().numpy()


@pytest.mark.parametrize("precision", ["bf16x3", "f16f6", "f32"])
def test_full_batch_determinism_and_batch_invariance(precision):
    from tf_kaldi_speaker_amd import synth
    tr, weights, params = _make(precision)
    utts = synth.synth_features(256, 300, 30, seed=1234)
    a = _run(tr, utts)
    b = _run(tr, utts)
    assert a.shape == (256, 512) and np.isfinite(a).all()
    assert np.array_equal(a, b)                                        # run-to-run bit identical
    perm = np.random.RandomState(0).permutation(256)
    c = _run(tr, [utts[i] for i in perm])
    rel = np.linalg.norm(c - a[perm], axis=1) / np.linalg.norm(a, axis=1)
    assert rel.max() <= 1e-6, rel.max()                                # position in the pack: fp32 reassociation only
    small = _run(tr, utts[100:103])
    rel = np.linalg.norm(small - a[100:103], axis=1) / np.linalg.norm(small, axis=1)
    assert rel.max() <= 1e-6, rel.max()                                # so is the rest of the batch
    f_all = _run(tr, utts[:8], node="tdnn4_relu").reshape(8, 286, 512)
    f_perm = _run(tr, utts[7::-1], node="tdnn4_relu").reshape(8, 286, 512)
    assert np.array_equal(f_perm[::-1], f_all)                         # frame-level rows: bit identical
    tr.close()


def test_ragged_pack_invariance_and_precision_ladder():
    from oracle import ref_numpy
    from tf_kaldi_speaker_amd import synth
    lens = np.random.RandomState(2024).randint(200, 1001, size=96)
    utts = synth.synth_features(len(lens), lens, 30, seed=77)
    tr16, weights, params = _make("bf16x3")
    tr32, _, _ = _make("f32")
    e16 = _run(tr16, utts)
    e32 = _run(tr32, utts)
    rel = np.linalg.norm(e16 - e32, axis=1) / np.linalg.norm(e32, axis=1)
    assert rel.max() <= 2e-5, rel.max()
    for i in (0, 41, 95):                                              # oracle on a sample (seconds each)
        ref = ref_numpy.predict(utts[i], weights, params, 30)
        assert np.linalg.norm(e16[i] - ref) / np.linalg.norm(ref) <= 1e-4
        assert np.linalg.norm(e32[i] - ref) / np.linalg.norm(ref) <= 1e-5
    alone = _run(tr16, [utts[41]])
    assert np.linalg.norm(alone[0] - e16[41]) / np.linalg.norm(e16[41]) <= 1e-6
    tr16.close()
    tr32.close()


def test_long_utterance_chunking_equals_weighted_mean():
    """extract.py:68-86 on T=2500 with S=1000: chunks [0,1000) [500,1500) [1000,2000) [1500,2500)."""
    from tf_kaldi_speaker_amd import extract, synth
    tr, _, _ = _make("bf16x3")
    long_utt = synth.synth_features(1, 2500, 30, seed=5)[0]
    out = []
    extract.extract_stream(tr.predict_list, iter([("k", long_utt)]), lambda k, v: out.append(v),
                           min_chunk_size=25, chunk_size=1000, normalize=True, batch_frames=76800)
    parts = extract.split_chunks(2500, 1000)
    assert parts == [(0, 1000), (500, 1000), (1000, 1000), (1500, 1000)]
    embs = _run(tr, [long_utt[s:s + n] for s, n in parts]).astype(np.float64)
    embs /= np.linalg.norm(embs, axis=1, keepdims=True)
    w = np.array([n for _, n in parts], dtype=np.float64)
    ref = (embs * w[:, None]).sum(0) / w.sum()
    ref /= np.linalg.norm(ref)
    assert np.linalg.norm(out[0] - ref) / np.linalg.norm(ref) <= 1e-6
    tr.close()


def test_extreme_geometries_many_tiny_and_one_huge_utterance():
    """The two ends of what a launcher batch can look like (extract.py:65-86: minimum length 25 by default, chunks of up to
    10 000 frames): 3 000 utterances of 15-40 frames (15 = the network's context + 1: one output frame; row maps and pooling
    segments dominate, every 64-row pooling pass spans several utterances) and one utterance of 12 000 frames next to a
    15-frame one (a pooling segment of 187 passes).  Split path against the exact path everywhere, the oracle on a sample."""
    from oracle import ref_numpy
    from tf_kaldi_speaker_amd import synth
    rs = np.random.RandomState(9)
    tiny = [int(t) for t in rs.randint(15, 41, size=3000)]
    tr16, weights, params = _make("bf16x3")
    tr32, _, _ = _make("f32")
    for lens, sample in ((tiny, (0, 1499, 2999)), ([12000, 15], (1,))):
        utts = synth.synth_features(len(lens), lens, 30, seed=78)
        e16, e32 = _run(tr16, utts), _run(tr32, utts)
        assert np.isfinite(e16).all() and e16.shape == (len(lens), 512)
        rel = np.linalg.norm(e16 - e32, axis=1) / np.linalg.norm(e32, axis=1)
        assert rel.max() <= 2e-5, rel.max()
        for i in sample:
            ref = ref_numpy.predict(utts[i], weights, params, 30)
            assert np.linalg.norm(e16[i] - ref) / np.linalg.norm(ref) <= 1e-4
        again = _run(tr16, utts)
        assert np.array_equal(again, e16)
    tr16.close()
    tr32.close()


def test_attention_and_resnet_batch_invariance():
    from tf_kaldi_speaker_amd import synth
    tr, _, _ = _make("bf16x3", synth.TDNN_ATT_PARAMS)
    utts = synth.synth_features(24, [200 + 13 * i for i in range(24)], 30, seed=3)
    a = _run(tr, utts)
    assert np.array_equal(_run(tr, utts), a)
    sub = _run(tr, utts[5:9])
    assert (np.linalg.norm(sub - a[5:9], axis=1) / np.linalg.norm(sub, axis=1)).max() <= 1e-6
    tr.close()
    from tf_kaldi_speaker_amd.params import Params
    from tf_kaldi_speaker_amd.trainer import Trainer
    p = dict(synth.RESNET_PARAMS)
    w = synth.synth_resnet_weights(p, seed=0)
    rn = Trainer(Params(**p), None, 40, single_cpu=True, device=0, precision="bf16x3")
    rn.build("predict")
    rn.load_weights(w)
    ru = synth.synth_features(6, [60, 75, 33, 120, 48, 90], 40, seed=4)
    r = _run(rn, ru)
    assert np.array_equal(_run(rn, ru), r)
    sub = _run(rn, ru[2:4])
    assert (np.linalg.norm(sub - r[2:4], axis=1) / np.linalg.norm(sub, axis=1)).max() <= 1e-6
    rn.close()
