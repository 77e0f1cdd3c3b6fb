#!/usr/bin/env python
"""In-process A/B of library options (xv_set_option): interleaved rounds of K forwards per setting, median ms per forward.
usage: python tools/ab_options.py <option> [tdnn|att|etdnn|resnet] [precision]     e.g.  python tools/ab_options.py slab3 att"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tf_kaldi_speaker_amd import synth  # noqa: E402
from tf_kaldi_speaker_amd.params import Params  # noqa: E402
from tf_kaldi_speaker_amd.trainer import Trainer  # noqa: E402

opt = sys.argv[1]
net = sys.argv[2] if len(sys.argv) > 2 else "tdnn"
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
if net == "resnet":
    params, dim, B, T = Params(**dict(synth.RESNET_PARAMS)), 40, 64, 300
    weights = synth.synth_resnet_weights(params, seed=0)
else:
    base = dict(synth.TDNN_ATT_PARAMS if net == "att" else synth.TDNN_STAT_PARAMS)
    if net == "etdnn":
        base.update(network_type="extended_tdnn", embedding_node="tdnn12_dense")
    params = Params(**base)
    dim, B, T = 30, 256, 300
    weights = synth.synth_weights(params, dim, seed=0)
feats = torch.from_numpy(np.concatenate(synth.synth_features(B, T, dim, seed=3))).cuda()
offs = np.arange(B + 1, dtype=np.int32) * T
trs = {}
for v in (1, 0):
    tr = Trainer(params, None, dim, single_cpu=True, device=0, precision=prec)
    tr.build("predict")
    tr.load_weights(weights)
    tr.set_option(opt, v)
    out = tr.predict_packed(feats, offs)
    trs[v] = (tr, out)
torch.cuda.synchronize()
K, rounds = 20, 7
res = {0: [], 1: []}
for r in range(rounds):
    for v in (1, 0):
        tr, out = trs[v]
        for _ in range(3):
            tr.predict_packed(feats, offs, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            tr.predict_packed(feats, offs, out=out)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / K * 1e3)
for v in (1, 0):
    print("%s=%d  %s %s: median %.4f ms  min %.4f  (%d rounds x %d forwards)  %.0f utt/s" %
          (opt, v, net, prec, np.median(res[v]), min(res[v]), rounds, K, B / np.median(res[v]) * 1e3))
print("same bits:", bool(torch.equal(trs[0][1], trs[1][1])))
