#!/usr/bin/env python
"""What does each stream-ordered piece of the pipelined host path cost the DEVICE per batch?  (no host work in the loop: everything
is enqueued ahead, one synchronisation at the end)   A forward only | B + result D2H + flag read-out | C + H2D of the next batch on a
copy stream | D the same H2D on the compute stream | E as C, H2D from a numpy view of the pinned buffer (what the CLI passes) |
F the product path: the last kernel writes the embeddings into the pinned slot, flags by one snapshot kernel (no H2D here)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tf_kaldi_speaker_amd import synth
from tf_kaldi_speaker_amd.params import Params
from tf_kaldi_speaker_amd.trainer import Trainer

params = Params(**dict(synth.TDNN_STAT_PARAMS))
tr = Trainer(params, None, 30, single_cpu=True, device=0)
tr.build("predict"); tr.load_weights(synth.synth_weights(params, 30, seed=0))
host = torch.from_numpy(np.concatenate(synth.synth_features(256, 300, 30, seed=1))).pin_memory()
view = torch.from_numpy(host.numpy())
print("pinned?", host.is_pinned(), "numpy view pinned?", view.is_pinned())
offs = np.arange(257, dtype=np.int32) * 300
devs = [host.cuda() for _ in range(3)]
out = tr.predict_packed(devs[0], offs)
pins = [torch.empty_like(out, device="cpu").pin_memory() for _ in range(3)]
flags = [torch.zeros(2, dtype=torch.int32).pin_memory() for _ in range(3)]
comp = torch.cuda.current_stream(); copy_s = torch.cuda.Stream()
N = 60
def run(mode):
    for _ in range(5): tr.predict_packed(devs[0], offs, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(N):
        k = i % 3
        if mode in "CE" or mode == "F":
            with torch.cuda.stream(copy_s):
                devs[k].copy_(view if mode == "E" else host, non_blocking=True)
                e = torch.cuda.Event(); e.record(copy_s)
            comp.wait_event(e)
        elif mode == "D":
            devs[k].copy_(host, non_blocking=True)
        if mode in "JKL":
            if mode == "J": tr.predict_packed(devs[k], offs, out=out); tr.flags_async(flags[k])        # flags kernel only
            if mode == "K": tr.predict_packed(devs[k], offs, out=pins[k])                              # direct pinned output only
            if mode == "L": tr.predict_packed(devs[k], offs, out=out); pins[k].copy_(out, non_blocking=True)   # D2H copy only
            continue
        if mode == "H":                                   # forward + an event record only
            tr.predict_packed(devs[k], offs, out=out)
            ev = torch.cuda.Event(); ev.record(comp)
            continue
        if mode == "G":                                   # result copy + flags only after every second forward
            tr.predict_packed(devs[k], offs, out=out)
            if i & 1:
                pins[k].copy_(out, non_blocking=True)
                tr.flags_async(flags[k])
            continue
        if mode == "F":
            tr.predict_packed(devs[k], offs, out=pins[k])
            tr.flags_async(flags[k])
            continue
        tr.predict_packed(devs[k], offs, out=out)
        if mode != "A":
            pins[k].copy_(out, non_blocking=True)
            tr.flags_async(flags[k])
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3, t_enq / N * 1e3
for rep in range(2):
    for mode in "AJKLF":
        ms, enq = run(mode)
        print("mode %s: %.3f ms per batch on the device (host enqueue %.3f ms)" % (mode, ms, enq))
# the same with 512 utterances per forward (one host-visible point per 2.5 ms instead of per 1.25 ms)
host2 = torch.cat([host, host]).pin_memory(); offs2 = np.arange(513, dtype=np.int32) * 300
dev2 = [host2.cuda() for _ in range(3)]
out2 = tr.predict_packed(dev2[0], offs2); pins2 = [torch.empty_like(out2, device="cpu").pin_memory() for _ in range(3)]
for mode in "AF":
    for _ in range(5): tr.predict_packed(dev2[0], offs2, out=out2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(N):
        k = i % 3
        if mode == "F":
            with torch.cuda.stream(copy_s):
                dev2[k].copy_(host2, non_blocking=True); e = torch.cuda.Event(); e.record(copy_s)
            comp.wait_event(e)
            tr.predict_packed(dev2[k], offs2, out=pins2[k]); tr.flags_async(flags[k])
        else:
            tr.predict_packed(dev2[k], offs2, out=out2)
    torch.cuda.synchronize()
    print("512 utterances, mode %s: %.3f ms per batch = %.0f utt/s" % (mode, (time.perf_counter() - t0) / N * 1e3, 512 * N / (time.perf_counter() - t0)))
tr.close()
