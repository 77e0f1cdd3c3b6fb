"""Angle between each utterance's `output` activation and the softmax weight vector of its speaker.

Counterpart of egs/voxceleb/v1/nnet/lib/extract_angle.py (same positional arguments): the class weights are
the transposed `softmax/output/kernel` of the checkpoint (:64-67), the embedding node is forced to `output`
(:70), the rspecifier must be an scp (:73-75), short utterances are skipped (:92-94), long ones truncated to
--chunk-size frames (:95-96), and `key angle` lines are written with angle = arccos of the cosine (:98-100).
Differences: the feature dimension comes from nnet/feature_dim like the other drivers (the reference asks the
training data directory), and --init (random weights) is not supported: there is nothing to analyse."""
import argparse
import logging
import os
import sys

import numpy as np

from . import model_io
from .kaldi_io import read_mat_scp
from .params import Params

log = logging.getLogger("xvec.extract_angle")


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("-g", "--gpu", type=int, default=-1, help="The GPU id (-1: LOCAL_RANK or 0; there is no CPU path).")
    parser.add_argument("-m", "--min-chunk-size", type=int, default=25,
                        help="The minimum length of the segments. Any segment shorted than this value will be ignored.")
    parser.add_argument("-s", "--chunk-size", type=int, default=10000, help="Longer utterances are truncated to this many frames.")
    parser.add_argument("--precision", type=str, default="", help="f32 | bf16x3 (extension)")
    parser.add_argument("model_dir", type=str, help="The model directory.")
    parser.add_argument("rspecifier", type=str, help="Kaldi feature scp file.")
    parser.add_argument("utt2spk", type=str, help="utt2spk")
    parser.add_argument("spklist", type=str, help="spklist")
    parser.add_argument("angles", type=str, help="output angles")
    return parser


def angle(weight, output):
    """extract_angle.py:98-99"""
    c = np.dot(weight, np.transpose(output)) / (np.sqrt(np.sum(weight ** 2)) * np.sqrt(np.sum(output ** 2)))
    return np.arccos(c)


def main(argv=None):
    args = build_parser().parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    nnet_dir = os.path.join(args.model_dir, "nnet")
    config_json = os.path.join(args.model_dir, "nnet/config.json")
    if not os.path.isfile(config_json):
        sys.exit("Cannot find params.json in %s" % config_json)
    params = Params(config_json)
    weights, _ = model_io.load_weights(nnet_dir)
    if weights is None or model_io.SOFTMAX_KERNEL not in weights:
        sys.exit("Cannot find %s in the checkpoint of %s" % (model_io.SOFTMAX_KERNEL, args.model_dir))
    class_weights = np.transpose(np.asarray(weights[model_io.SOFTMAX_KERNEL], dtype=np.float32))   # [speakers, E]
    params.embedding_node = "output"                                                # extract_angle.py:70
    with open(os.path.join(nnet_dir, "feature_dim"), "r") as f:
        dim = int(f.readline().strip())
    if "selected_dim" in params.dict:
        dim = params.selected_dim
    if args.rspecifier.rsplit(".", 1)[-1] != "scp":
        sys.exit("The rspecifier must be scp")
    spk2int = {}
    with open(args.spklist, "r") as f:
        for line in f:
            spk, i = line.strip().split(" ")
            spk2int[spk] = int(i)
    utt2spk = {}
    with open(args.utt2spk, "r") as f:
        for line in f:
            utt, spk = line.strip().split(" ")
            utt2spk[utt] = spk
    from .trainer import Trainer
    trainer = Trainer(params, args.model_dir, dim, num_speakers=class_weights.shape[0], single_cpu=True,
                      device=args.gpu if args.gpu >= 0 else None, precision=args.precision or None)
    trainer.build("predict")
    n = 0
    with open(args.angles, "w") as fp_out:
        for key, feature in read_mat_scp(args.rspecifier):
            i = spk2int[utt2spk[key]]
            if feature.shape[0] < args.min_chunk_size:
                log.info("[INFO] Key %s length too short, %d < %d, skip." % (key, feature.shape[0], args.min_chunk_size))
                continue
            if feature.shape[0] > args.chunk_size:
                feature = feature[:args.chunk_size]
            output = trainer.predict(feature)
            fp_out.write("%s %f\n" % (key, angle(class_weights[i, :], output)))
            n += 1
    trainer.close()
    log.info("Wrote %d angles." % n)
    return 0


if __name__ == "__main__":
    sys.exit(main())
