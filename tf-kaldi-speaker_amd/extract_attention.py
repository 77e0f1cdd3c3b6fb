"""Attention-weight dump: Kaldi matrix ark in -> matrix ark out ([heads, frames'] per utterance).

Counterpart of egs/voxceleb/v1/nnet/lib/extract_attention.py (same command line): the embedding
node is forced to `attention_weights` (:41), short utterances are skipped (:53-55), long ones are
truncated to --chunk-size frames (:56-57), and the [h, l] softmax weights of model/pooling.py:197-198
are written with write_mat (:58-59)."""
import argparse
import logging
import os
import sys

import numpy as np

from .kaldi_io import open_or_fd, read_mat_ark, write_mat
from .params import Params

log = logging.getLogger("xvec.extract_attention")


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("-g", "--gpu", type=int, default=-1, help="The GPU id (-1: LOCAL_RANK or 0; there is no CPU path).")
    parser.add_argument("-m", "--min-chunk-size", type=int, default=25,
                        help="The minimum length of the segments. Any segment shorted than this value will be ignored.")
    parser.add_argument("-s", "--chunk-size", type=int, default=10000, help="Longer utterances are truncated to this many frames.")
    parser.add_argument("--precision", type=str, default="", help="f32 | bf16x3 (extension)")
    parser.add_argument("model_dir", type=str, help="The model directory.")
    parser.add_argument("rspecifier", type=str, help="Kaldi feature rspecifier (or ark file).")
    parser.add_argument("wspecifier", type=str, help="Kaldi output wspecifier (or ark file).")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    nnet_dir = os.path.join(args.model_dir, "nnet")
    config_json = os.path.join(args.model_dir, "nnet/config.json")
    if not os.path.isfile(config_json):
        sys.exit("Cannot find params.json in %s" % config_json)
    params = Params(config_json)
    params.embedding_node = "attention_weights"                                    # extract_attention.py:41
    with open(os.path.join(nnet_dir, "feature_dim"), "r") as f:
        dim = int(f.readline().strip())
    from .trainer import Trainer
    trainer = Trainer(params, args.model_dir, dim, single_cpu=True, device=args.gpu if args.gpu >= 0 else None,
                      precision=args.precision or None)
    trainer.build("predict")
    if args.rspecifier.rsplit(".", 1)[-1] == "scp":
        sys.exit("The rspecifier must be ark or input pipe")
    fp_out = open_or_fd(args.wspecifier, "wb")
    n = 0
    for key, feature in read_mat_ark(args.rspecifier):
        if feature.shape[0] < args.min_chunk_size:
            log.info("[INFO] Key %s length too short, %d < %d, skip." % (key, feature.shape[0], args.min_chunk_size))
            continue
        if feature.shape[0] > args.chunk_size:
            feature = feature[:args.chunk_size]
        write_mat(fp_out, np.ascontiguousarray(trainer.predict(feature), dtype=np.float32), key=key)
        n += 1
    fp_out.close()
    proc = getattr(fp_out, "_xv_proc", None)
    if proc is not None:
        proc.wait()
    trainer.close()
    log.info("Wrote attention weights of %d utterances." % n)
    return 0


if __name__ == "__main__":
    sys.exit(main())
