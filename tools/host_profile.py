#!/usr/bin/env python
"""Where does the host time of one pipelined batch go?  Times the pieces of Trainer.submit_list / collect on the bench batch
(256 x [300, 30]) and prints the driver loop's own wait accounting for an ark of n utterances.  usage: host_profile.py [n_utts]"""
import logging, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tf_kaldi_speaker_amd import synth, extract, kaldi_io, model_io
from tf_kaldi_speaker_amd.params import Params
from tf_kaldi_speaker_amd.trainer import Trainer

n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
params = Params(**dict(synth.TDNN_STAT_PARAMS))
weights = synth.synth_weights(params, 30, seed=0)
tr = Trainer(params, None, 30, single_cpu=True, device=0)
tr.build("predict"); tr.load_weights(weights)
utts = synth.synth_features(256, 300, 30, seed=1)
for _ in range(5): tr.predict_list(utts)
lens = [300] * 256
sl_pin = torch.empty((76800, 30), dtype=torch.float32, pin_memory=True)
t = []
for _ in range(20):
    t0 = time.perf_counter(); tr._pack(utts, lens, 30, sl_pin); t.append(time.perf_counter() - t0)
print("pack 256 x [300,30] into pinned staging: %.3f ms (min of 20)" % (min(t) * 1e3))
for depth in (1, 2, 3):
    torch.cuda.synchronize()
    tickets, sub, col = [], 0.0, 0.0
    t_all = time.perf_counter()
    for i in range(40):
        t0 = time.perf_counter(); tickets.append(tr.submit_list(utts)); sub += time.perf_counter() - t0
        if len(tickets) == depth:
            t0 = time.perf_counter(); tr.collect(tickets.pop(0)); col += time.perf_counter() - t0
    while tickets:
        t0 = time.perf_counter(); tr.collect(tickets.pop(0)); col += time.perf_counter() - t0
    el = time.perf_counter() - t_all
    print("depth %d: %.0f utt/s; per batch %.3f ms = submit %.3f + collect %.3f" % (depth, 40 * 256 / el, el / 40 * 1e3, sub / 40 * 1e3, col / 40 * 1e3))
dev = torch.from_numpy(np.concatenate(utts)).cuda(); offs = np.arange(257, dtype=np.int32) * 300
out = tr.predict_packed(dev, offs); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40): tr.predict_packed(dev, offs, out=out)
t_enq = time.perf_counter() - t0; torch.cuda.synchronize(); t_tot = time.perf_counter() - t0
print("predict_packed: host enqueue %.3f ms per forward, device %.3f ms" % (t_enq / 40 * 1e3, t_tot / 40 * 1e3))
tr.close()
tmp = tempfile.mkdtemp(prefix="xvhp_", dir="/tmp")
model_io.save_model(os.path.join(tmp, "exp"), dict(params.dict), 30, weights, step=1)
ark = os.path.join(tmp, "feats.ark")
base = np.random.RandomState(0).standard_normal((364, 30)).astype(np.float32)
with open(ark, "wb") as f:
    for i in range(n_utts): kaldi_io.write_mat(f, base[i % 64:i % 64 + 300], key="utt%07d" % i)
logging.basicConfig(level=logging.INFO, format="%(message)s")
logging.getLogger("xvec.extract").setLevel(logging.INFO)
class Only(logging.Filter):
    def filter(self, r): return "driver loop" in r.getMessage() or r.getMessage().startswith("Extracted")
for h in logging.getLogger().handlers: h.addFilter(Only())
for rep in range(2):
    extract.main(["--gpu", "0", os.path.join(tmp, "exp"), "ark:" + ark, "ark:" + os.path.join(tmp, "xv.ark")])
