import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_kaldi_speaker_amd import synth
from tf_kaldi_speaker_amd.params import Params
from tf_kaldi_speaker_amd.trainer import Trainer
params = Params(**dict(synth.TDNN_STAT_PARAMS))
weights = synth.synth_weights(params, 30, seed=0)
trs = {}
for prec in ("f32", "f16f6"):
    tr = Trainer(params, None, 30, single_cpu=True, device=0, precision=prec)
    tr.build("predict"); tr.load_weights(weights); trs[prec] = tr
worst = 0.0
for T in (15, 16, 17, 29, 30, 127, 128, 129, 141, 142, 143, 270, 300, 1000, 4097):
    u = synth.synth_features(1, T, 30, seed=T)
    trs["f16f6"]._ws is not None and trs["f16f6"]._ws.fill_(255)
    a = trs["f16f6"].predict(u[0]).astype(np.float64)
    b = trs["f32"].predict(u[0]).astype(np.float64)
    e = np.linalg.norm(a - b) / np.linalg.norm(b)
    worst = max(worst, e)
    print("T %5d  rel vs exact %.2e  finite %s" % (T, e, np.isfinite(a).all()))
    assert np.isfinite(a).all() and e <= 1e-4
print("OK worst %.2e" % worst)
