#!/usr/bin/env python
"""Per-layer hipEvent times of the default forward, several rounds (quick look while tuning).  usage: layer_times.py [tdnn|att|etdnn|resnet] [precision]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_kaldi_speaker_amd import synth
from tf_kaldi_speaker_amd.params import Params
from tf_kaldi_speaker_amd.trainer import Trainer
net = sys.argv[1] if len(sys.argv) > 1 else "tdnn"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
if net == "resnet":
    params, dim, B = Params(**dict(synth.RESNET_PARAMS)), 40, 64
    weights = synth.synth_resnet_weights(params, seed=0)
else:
    base = dict(synth.TDNN_ATT_PARAMS if net == "att" else synth.TDNN_STAT_PARAMS)
    if net == "etdnn":
        base.update(network_type="extended_tdnn", embedding_node="tdnn12_dense")
    params, dim, B = Params(**base), 30, 256
    weights = synth.synth_weights(params, dim, seed=0)
feats = torch.from_numpy(np.concatenate(synth.synth_features(B, 300, dim, seed=3))).cuda()
offs = np.arange(B + 1, dtype=np.int32) * 300
tr = Trainer(params, None, dim, single_cpu=True, device=0, precision=prec)
tr.build("predict"); tr.load_weights(weights)
out = tr.predict_packed(feats, offs)
for rnd in range(4):
    for _ in range(3): tr.predict_packed(feats, offs, out=out)
    torch.cuda.synchronize()
    tr.profile_begin(8192)
    for _ in range(20): tr.predict_packed(feats, offs, out=out)
    recs, _ = tr.profile_end()
    print("total %.4f ms | " % sum(r["ms"] for r in recs) + " ".join("%s %.4f" % (r["name"][:12], r["ms"]) for r in recs[:int(os.environ.get("LT_MAX", "14"))]))
