"""Experiment: consecutive batches on alternating HIP streams (two trainers = two workspaces), so that the small
latency-bound kernels and the last tile rounds of one forward overlap the big GEMMs of the next.
usage: two_streams.py [tdnn|att|resnet] [nstreams]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_kaldi_speaker_amd import synth
from tf_kaldi_speaker_amd.params import Params
from tf_kaldi_speaker_amd.trainer import Trainer
net = sys.argv[1] if len(sys.argv) > 1 else "tdnn"
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if net == "resnet":
    params, dim, B = Params(**dict(synth.RESNET_PARAMS)), 40, 64
    weights = synth.synth_resnet_weights(params, seed=0)
else:
    params, dim, B = Params(**dict(synth.TDNN_ATT_PARAMS if net == "att" else synth.TDNN_STAT_PARAMS)), 30, 256
    weights = synth.synth_weights(params, dim, seed=0)
feats = torch.from_numpy(np.concatenate(synth.synth_features(B, 300, dim, seed=3))).cuda()
offs = np.arange(B + 1, dtype=np.int32) * 300
trs, streams, outs = [], [], []
for i in range(ns):
    tr = Trainer(params, None, dim, single_cpu=True, device=0, precision="bf16x3")
    tr.build("predict"); tr.load_weights(weights)
    trs.append(tr); streams.append(torch.cuda.Stream()); outs.append(tr.predict_packed(feats, offs))
torch.cuda.synchronize()
def run(n, k):
    for j in range(n):
        i = j % k
        with torch.cuda.stream(streams[i]):
            trs[i].predict_packed(feats, offs, out=outs[i])
for k in (1, ns, 1, ns):
    run(10, k); torch.cuda.synchronize()
    rates = []
    for rep in range(5):
        t0 = time.perf_counter(); run(40, k); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        rates.append(40 * B / dt)
    print("%s streams=%d: median %.0f utt/s (min %.0f max %.0f)" % (net, k, sorted(rates)[2], min(rates), max(rates)))
assert torch.equal(outs[0], outs[-1])
