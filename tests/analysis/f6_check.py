"""Quick look at the f16f6 precision: conv-layer outputs and the x-vector against the exact fp32 path and the float64 oracle,
next to f16x3 (ragged batch incl. a 15-frame utterance).  usage: python tests/analysis/f6_check.py"""
import os
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_kaldi_speaker_amd import synth
from tf_kaldi_speaker_amd.params import Params
from tf_kaldi_speaker_amd.trainer import Trainer
from oracle import ref_numpy
params = Params(**dict(synth.TDNN_STAT_PARAMS))
weights = synth.synth_weights(params, 30, seed=0)
lens = [300, 64, 200, 15, 129, 333]
utts = synth.synth_features(len(lens), lens, 30, seed=7)
feats = torch.from_numpy(np.concatenate(utts)).cuda()
offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
res = {}
for prec in ("f32", "f16x3", "f16f6"):
    tr = Trainer(params, None, 30, single_cpu=True, device=0, precision=prec)
    tr.build("predict"); tr.load_weights(weights)
    res[prec] = {n: tr.predict_packed(feats, offs, n).cpu().numpy().astype(np.float64) for n in ("tdnn2_relu", "tdnn3_relu", "tdnn6_dense")}
    tr.close()
for n in ("tdnn2_relu", "tdnn3_relu", "tdnn6_dense"):
    for prec in ("f16x3", "f16f6"):
        d = np.linalg.norm(res[prec][n] - res["f32"][n]) / np.linalg.norm(res["f32"][n])
        print(n, prec, "rel vs f32 path %.3e" % d)
for i in (0, 3, 5):
    ref = ref_numpy.predict(utts[i], weights, params, 30)
    print("utt", i, "f16f6 vs oracle %.3e" % (np.linalg.norm(res["f16f6"]["tdnn6_dense"][i] - ref) / np.linalg.norm(ref)))
