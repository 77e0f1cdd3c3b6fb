#!/usr/bin/env python
"""Where does the bf16x3 error of `attention_weights` come from?  numpy emulation of the split arithmetic (hi = rn_bf16(x),
lo = rn_bf16(x - hi); a*b ~ ah*bh + ah*bl + al*bh, sums in float64) on the peaky-query case of
tests/test_gpu_parity.py::test_every_endpoint_self_attention, one layer at a time.  CPU only.
Output of the committed version: all layers x3 8.9e-5 (GPU: 7.0e-5); only key1 4.2e-5, only tdnn4 2.7e-5, every other
single layer ~1.3e-5; key0+key1 exact 3.9e-5; a fourth lo*lo product everywhere 4.5e-5; fp32-rounded inputs 3.5e-7."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref_numpy
from tf_kaldi_speaker_amd import synth

def bf16(x):
    x = np.asarray(x, np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).astype(np.float64)

def split(x):
    x32 = np.asarray(x, np.float32)      # activations are fp32 on the GPU
    hi = bf16(x32)
    lo = bf16((x32.astype(np.float64) - hi).astype(np.float32))
    return hi, lo

MODE = {}
def mm(a, w, name):
    m = MODE.get(name, MODE.get('*', 'x3'))
    if m == 'exact':
        return a @ np.asarray(w, np.float64)
    ah, al = split(a); wh, wl = split(w)
    y = ah @ wh + ah @ wl + al @ wh
    if m == 'x4':
        y = y + al @ wl
    if m == 'f32in':       # exact products of fp32-rounded inputs
        return np.asarray(a, np.float32).astype(np.float64) @ np.asarray(w, np.float32).astype(np.float64)
    return y

cur = ['']
orig_conv, orig_dense = ref_numpy.conv_valid, ref_numpy.dense_layer
counter = [0]
names_conv = ['tdnn1', 'tdnn2', 'tdnn3']
def conv_emu(x, kernel, bias):
    name = names_conv[counter[0] % 3]; counter[0] += 1
    kernel = np.asarray(kernel, np.float64)
    k = kernel.shape[1]; b, l, c = x.shape; lo = l - k + 1
    y = np.zeros((b, lo, kernel.shape[3]))
    for j in range(k):
        y += mm(x[:, j:j + lo, :], kernel[0, j], name)
    return y + np.asarray(bias, np.float64)
dn = [0]
names_dense = ['tdnn4', 'tdnn5', 'key0', 'key1', 'tdnn6', 'tdnn7']
def dense_emu(x, kernel, bias):
    name = names_dense[dn[0] % 6]; dn[0] += 1
    return mm(x, kernel, name) + np.asarray(bias, np.float64)

params = dict(synth.TDNN_ATT_PARAMS)
weights = synth.synth_weights(params, 30, seed=2)
weights["tdnn/attention/query"] = weights["tdnn/attention/query"] * 1000.0
feats = np.stack(synth.synth_features(2, 52, 30, seed=12))
_, ep = ref_numpy.entire_network(feats, weights, params)
ref_w = ep["attention_weights"]
print("max weight", ref_w.max())
ref_numpy.conv_valid, ref_numpy.dense_layer = conv_emu, dense_emu
def run(mode):
    MODE.clear(); MODE.update(mode); counter[0] = 0; dn[0] = 0
    _, e = ref_numpy.entire_network(feats, weights, params)
    w = e["attention_weights"]
    return np.linalg.norm(w - ref_w) / np.linalg.norm(ref_w), np.linalg.norm(e["tdnn6_dense"] - ep["tdnn6_dense"]) / np.linalg.norm(ep["tdnn6_dense"])
print("all x3          ", run({'*': 'x3'}))
print("all exact (sanity)", run({'*': 'exact'}))
for only in ['tdnn1', 'tdnn2', 'tdnn3', 'tdnn4', 'key0', 'key1']:
    print("only %s x3" % only, run({'*': 'exact', only: 'x3'}))
print("key0,key1 x4    ", run({'*': 'x3', 'key0': 'x4', 'key1': 'x4'}))
print("all x4          ", run({'*': 'x4'}))
print("key1 exact      ", run({'*': 'x3', 'key1': 'exact'}))
print("key0+key1 exact ", run({'*': 'x3', 'key0': 'exact', 'key1': 'exact'}))
print("all f32in       ", run({'*': 'f32in'}))
