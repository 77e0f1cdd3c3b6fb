"""Frame-level embedding dump: Kaldi matrix ark in -> matrix ark out, one row per input frame.

Counterpart of egs/voxceleb/v1/nnet/lib/extract_frame.py (same command line): utterances shorter
than --min-chunk-size are skipped (:64-66); longer than --chunk-size are cut into NON-overlapping
chunks (:67-76); the network's output (T - context rows) is padded back to T rows by repeating its
first and last row `pad = (T - T') // 2` times on each side (:78-91; `/ 2` there is Python-2 integer
division); the rows of all chunks are concatenated and written with write_mat (:93-94).
Chunks are packed into ragged device batches; output order is input order."""
import argparse
import logging
import os
import sys

import numpy as np

from .kaldi_io import open_or_fd, read_mat_ark, write_mat
from .params import Params

log = logging.getLogger("xvec.extract_frame")


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("-g", "--gpu", type=int, default=-1, help="The GPU id (-1: LOCAL_RANK or 0; there is no CPU path).")
    parser.add_argument("-m", "--min-chunk-size", type=int, default=25,
                        help="The minimum length of the segments. Any segment shorted than this value will be ignored.")
    parser.add_argument("-s", "--chunk-size", type=int, default=10000,
                        help="Segments longer than this value are split (without overlap) before extraction.")
    parser.add_argument("--node", type=str, default="", help="The node to output the embeddings.")
    parser.add_argument("--batch-frames", type=int, default=76800, help="Frames per device batch (extension).")
    parser.add_argument("--precision", type=str, default="", help="f32 | bf16x3 (extension)")
    parser.add_argument("model_dir", type=str, help="The model directory.")
    parser.add_argument("rspecifier", type=str, help="Kaldi feature rspecifier (or ark file).")
    parser.add_argument("wspecifier", type=str, help="Kaldi output wspecifier (or ark file).")
    return parser


def split_plain(num_frames, chunk_size):
    """extract_frame.py:70-76: ceil(T/S) consecutive chunks, the last one shorter."""
    n = int(np.ceil(float(num_frames) / chunk_size))
    return [(i * chunk_size, min(chunk_size, num_frames - i * chunk_size)) for i in range(n)]


def pad_edges(emb, length):
    """extract_frame.py:78-91: repeat the first / last output row (length - len(emb)) // 2 times each."""
    pad = (length - emb.shape[0]) // 2
    return np.concatenate([np.tile(emb[0], [pad, 1]), emb, np.tile(emb[-1], [pad, 1])], axis=0)


def extract_frames_stream(embed_fn, items, write_fn, min_chunk_size=25, chunk_size=10000, batch_frames=76800):
    """`embed_fn(list of [T_i,d])` -> list of [T_i', E] frame-level outputs.  Returns (#written, #skipped)."""
    pending, pieces, frames = [], [], 0
    done = skipped = 0

    def flush():
        nonlocal pending, pieces, frames, done
        if not pending:
            return
        outs = embed_fn(pieces)
        for key, idx, lens, total in pending:
            emb = np.concatenate([pad_edges(np.asarray(outs[i]), n) for i, n in zip(idx, lens)], axis=0)
            assert emb.shape[0] == total                                      # extract_frame.py:93
            write_fn(key, np.ascontiguousarray(emb, dtype=np.float32))
            done += 1
        pending, pieces, frames = [], [], 0

    for key, feature in items:
        t = feature.shape[0]
        if t < min_chunk_size:
            log.info("[INFO] Key %s length too short, %d < %d, skip." % (key, t, min_chunk_size))
            skipped += 1
            continue
        parts = split_plain(t, chunk_size) if t > chunk_size else [(0, t)]
        idx = []
        for start, n in parts:
            idx.append(len(pieces))
            pieces.append(feature[start:start + n])
        pending.append((key, idx, [n for _, n in parts], t))
        frames += t
        if frames >= batch_frames:
            flush()
    flush()
    return done, skipped


def main(argv=None):
    args = build_parser().parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    nnet_dir = os.path.join(args.model_dir, "nnet")
    config_json = os.path.join(args.model_dir, "nnet/config.json")
    if not os.path.isfile(config_json):
        sys.exit("Cannot find params.json in %s" % config_json)
    params = Params(config_json)
    if len(args.node) != 0:
        params.embedding_node = args.node
    log.info("Extract embedding from %s" % params.embedding_node)
    with open(os.path.join(nnet_dir, "feature_dim"), "r") as f:
        dim = int(f.readline().strip())
    from .trainer import Trainer
    trainer = Trainer(params, args.model_dir, dim, single_cpu=True, device=args.gpu if args.gpu >= 0 else None,
                      precision=args.precision or None)
    trainer.build("predict")
    if args.rspecifier.rsplit(".", 1)[-1] == "scp":
        sys.exit("The rspecifier must be ark or input pipe")
    fp_out = open_or_fd(args.wspecifier, "wb")
    done, skipped = extract_frames_stream(trainer.predict_list, read_mat_ark(args.rspecifier),
                                          lambda key, m: write_mat(fp_out, m, key=key),
                                          args.min_chunk_size, args.chunk_size, args.batch_frames)
    fp_out.close()
    proc = getattr(fp_out, "_xv_proc", None)
    if proc is not None:
        proc.wait()
    trainer.close()
    log.info("Wrote %d matrices (%d utterances skipped)." % (done, skipped))
    return 0


if __name__ == "__main__":
    sys.exit(main())
