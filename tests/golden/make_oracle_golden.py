#!/usr/bin/env python
"""Regression goldens of the ORACLE itself (not a pin from the reference): embeddings that oracle/ref_numpy.py
(float64) produces for seeded synthetic models and one seeded 300-frame utterance, committed so that
 * a later change of the oracle or of the synthetic generators is noticed (tests/test_oracle.py), and
 * the GPU box has expected vectors that do not depend on running the oracle there (tests/test_gpu_parity.py).
Run from the repo root:  python tests/golden/make_oracle_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_numpy  # noqa: E402
from tf_kaldi_speaker_amd import synth  # noqa: E402

CASES = {     # name -> (params, feature dim, weight seed, feature seed, frames)
    "tdnn_stat": (dict(synth.TDNN_STAT_PARAMS), 30, 0, 1234, 300),
    "tdnn_att": (dict(synth.TDNN_ATT_PARAMS), 30, 0, 1234, 300),
    "etdnn_stat": (dict(synth.TDNN_STAT_PARAMS, network_type="extended_tdnn", embedding_node="tdnn12_dense"), 30, 0, 1234, 300),
    "resnet18": (dict(synth.RESNET_PARAMS), 40, 0, 1234, 120),
}


def compute(name):
    params, dim, wseed, fseed, frames = CASES[name]
    if params.get("network_type") == "resnet_18":
        weights = synth.synth_resnet_weights(params, seed=wseed)
    else:
        weights = synth.synth_weights(params, dim, seed=wseed)
    feats = synth.synth_features(1, frames, dim, seed=fseed)[0]
    return np.asarray(ref_numpy.predict(feats, weights, params, dim), dtype=np.float64)


def main():
    out = {name: compute(name) for name in CASES}
    np.savez(os.path.join(HERE, "oracle_embeddings.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, float(np.linalg.norm(v)))


if __name__ == "__main__":
    main()
