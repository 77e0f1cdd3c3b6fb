#!/usr/bin/env python
"""Where a job's start-up goes (fresh process): imports, first touch of the GPU, graph build, weight upload, pinned staging
buffers, first forward (tiny batch, full batch) and a steady one.  usage: python tools/startup_probe.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
t0 = time.perf_counter()
import numpy as np, torch
from tf_kaldi_speaker_amd import synth
from tf_kaldi_speaker_amd.params import Params
from tf_kaldi_speaker_amd.trainer import Trainer
t1 = time.perf_counter()
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t2 = time.perf_counter()
params = Params(**dict(synth.TDNN_STAT_PARAMS))
w = synth.synth_weights(params, 30, seed=0)
tr = Trainer(params, None, 30, single_cpu=True, device=0); tr.build("predict")
t3 = time.perf_counter()
tr.load_weights(w); torch.cuda.synchronize()
t4 = time.perf_counter()
pins = [torch.empty((153600 + 65536) * 64, dtype=torch.float32, pin_memory=True) for _ in range(6)]
t5 = time.perf_counter()
feats = torch.from_numpy(np.concatenate(synth.synth_features(512, 300, 30, seed=1))).cuda()
offs = np.arange(513, dtype=np.int32) * 300
torch.cuda.synchronize(); t6 = time.perf_counter()
tiny = feats[:60]; toffs = np.array([0, 30, 60], dtype=np.int32)
tr.predict_packed(tiny, toffs); torch.cuda.synchronize(); t7 = time.perf_counter()
tr.predict_packed(feats, offs); torch.cuda.synchronize(); t8 = time.perf_counter()
tr.predict_packed(feats, offs); torch.cuda.synchronize(); t9 = time.perf_counter()
print("imports %.2f | cuda init %.2f | build %.3f | load_weights %.2f | 6 pinned buffers (%.0f MB each) %.3f | synth+H2D %.2f | tiny first forward %.3f | first full forward %.3f | second %.4f"
      % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, pins[0].numel() * 4 / 1e6, t5 - t4, t6 - t5, t7 - t6, t8 - t7, t9 - t8))
