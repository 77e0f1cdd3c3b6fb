#!/usr/bin/env python
"""Stress the fast kernels (argument 3: f16f6 = the default precision, or bf16x3) over random batch geometries against the exact fp32-MFMA path of the same
library (itself checked against the oracle by tests/): embeddings must agree to 1e-4 relative L2, repeated runs of one
geometry must be bit-identical.  Product code only (no oracle).  usage: python tools/stress_shapes.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_kaldi_speaker_amd import synth  # noqa: E402
from tf_kaldi_speaker_amd.params import Params  # noqa: E402
from tf_kaldi_speaker_amd.trainer import Trainer  # noqa: E402


def make(params, dim, precision, weights):
    tr = Trainer(Params(**dict(params)), None, dim, single_cpu=True, device=0, precision=precision)
    tr.build("predict")
    tr.load_weights(weights)
    return tr


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    fast_precision = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
    rng = np.random.default_rng(seed)
    nets = {
        "tdnn_stat": (dict(synth.TDNN_STAT_PARAMS), 30, 15),
        "tdnn_att": (dict(synth.TDNN_ATT_PARAMS), 30, 15),
        "etdnn": (dict(synth.TDNN_STAT_PARAMS, network_type="extended_tdnn", embedding_node="tdnn12_dense"), 30, 23),
        "resnet18": (dict(synth.RESNET_PARAMS), 40, 1),
    }
    models = {}
    for name, (params, dim, tmin) in nets.items():
        w = synth.synth_resnet_weights(params, seed=1) if name == "resnet18" else synth.synth_weights(params, dim, seed=1)
        models[name] = (make(params, dim, fast_precision, w), make(params, dim, "f32", w), dim, tmin)
    worst, t0 = 0.0, time.time()
    for c in range(cases):
        name = list(nets)[c % len(nets)]
        fast, exact, dim, tmin = models[name]
        heavy = name == "resnet18"
        nutt = int(rng.integers(1, 20 if heavy else 300))
        hi = 200 if heavy else int(rng.choice([40, 130, 400, 1300]))
        lens = [int(x) for x in rng.integers(tmin, hi + 1, size=nutt)]
        utts = synth.synth_features(nutt, lens, dim, seed=1000 + c)
        a = np.asarray(fast.predict_list(utts), dtype=np.float64)
        a2 = np.asarray(fast.predict_list(utts), dtype=np.float64)
        b = np.asarray(exact.predict_list(utts), dtype=np.float64)
        assert np.array_equal(a, a2), ("non-deterministic", name, nutt)
        assert np.all(np.isfinite(a)), ("non-finite", name, nutt)
        err = float(np.max(np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)))
        worst = max(worst, err)
        print("case %3d %-9s utts %3d frames %6d  max rel-L2 vs f32 path %.2e" % (c, name, nutt, sum(lens), err), flush=True)
        assert err <= 1e-4, (name, nutt, err)
    print("OK: %d cases, worst %.2e, %.1f s" % (cases, worst, time.time() - t0))


if __name__ == "__main__":
    main()
