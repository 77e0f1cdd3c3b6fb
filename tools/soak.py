#!/usr/bin/env python
"""Soak: N forwards of the BASELINE geometry on the default kernels, output compared bit-for-bit with the first one
every `every` steps (a lost wait or a race in the hand-scheduled GEMM would show up as a flipped bit sooner or later).
usage: python tools/soak.py [steps] [every]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tf_kaldi_speaker_amd import synth  # noqa: E402
from tf_kaldi_speaker_amd.params import Params  # noqa: E402
from tf_kaldi_speaker_amd.trainer import Trainer  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    params = dict(synth.TDNN_STAT_PARAMS)
    precision = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
    tr = Trainer(Params(**params), None, 30, single_cpu=True, device=0, precision=precision)
    tr.build("predict")
    tr.load_weights(synth.synth_weights(params, 30, seed=0))
    B, T = 256, 300
    feats = torch.from_numpy(np.concatenate(synth.synth_features(B, T, 30, seed=5))).cuda()
    offs = np.arange(B + 1, dtype=np.int32) * T
    out = torch.empty((B, 512), dtype=torch.float32, device="cuda")
    ref = tr.predict_packed(feats, offs).clone()
    t0, bad = time.time(), 0
    for i in range(steps):
        tr.predict_packed(feats, offs, out=out)
        if (i + 1) % every == 0:
            if not torch.equal(out, ref):
                bad += 1
                print("step %d: output differs from the first forward (max abs %.3e)" % (i + 1, float((out - ref).abs().max())), flush=True)
        if (i + 1) % 2000 == 0:
            print("step %d  %.1f s  mismatches %d" % (i + 1, time.time() - t0, bad), flush=True)
    torch.cuda.synchronize()
    print("%s: %d forwards, %d checks, %d mismatches, %.1f s" % ("OK" if bad == 0 else "FAILED", steps, steps // every, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
