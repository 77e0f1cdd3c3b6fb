import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch
from tf_kaldi_speaker_amd import synth
from tf_kaldi_speaker_amd.params import Params
from tf_kaldi_speaker_amd.trainer import Trainer
params = Params(**dict(synth.TDNN_STAT_PARAMS))
w = synth.synth_weights(params, 30, seed=0)
tr = Trainer(params, None, 30, single_cpu=True, device=0, precision="bf16x3")
tr.build("predict"); tr.load_weights(w)
utts = synth.synth_features(256, 300, 30, seed=1)
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10): tr.predict_list(utts)
    print("predict_list x10: %.1f ms/call" % ((time.perf_counter() - t0) * 100))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): tr.predict_list(utts)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
