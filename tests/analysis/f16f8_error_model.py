#!/usr/bin/env python
"""Would a cheaper split meet the 1e-4 bar?  numpy emulation (CPU only) of
     a*b ~ f16(a)*f16(b)  +  q8(f16(a))*q8(b - f16(b))  +  q8(a - f16(a))*q8(f16(b))
with q8 = OCP fp8 e4m3 under a power-of-two scale per 32-element K block (what v_mfma_scale_f32_16x16x128_f8f6f4 consumes at twice the
bf16 rate): 1 + 2 x 1/2 = 2 MFMA units per product instead of the 3 of bf16x3 / f16x3.  TDNN x-vector (statistics pooling), 4 utterances
of 300 frames, sums in float64, against the exact float64 forward.  Also prints f16x3 and a one-sided variant for reference."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref_numpy
from tf_kaldi_speaker_amd import synth


def f16(x):
    return np.asarray(x, np.float64).astype(np.float16).astype(np.float64)


def q8_blocks(v, axis, mant=3, emin=-6, vmax=448.0):
    """e4m3 (mant=3, emin=-6, max 448) or e5m2 (mant=2, emin=-14, max 57344) with one power-of-two scale per 32 elements along `axis`."""
    v = np.moveaxis(np.asarray(v, np.float64), axis, -1)
    shp = v.shape
    k = shp[-1]
    pad = (-k) % 32
    if pad:
        v = np.concatenate([v, np.zeros(shp[:-1] + (pad,))], -1)
    b = v.reshape(shp[:-1] + (-1, 32))
    amax = np.abs(b).max(-1, keepdims=True)
    e = np.where(amax > 0, np.ceil(np.log2(np.maximum(amax, 1e-300) / vmax)), 0.0)
    s = 2.0 ** e
    x = b / s
    ax = np.abs(x)
    ex = np.floor(np.log2(np.maximum(ax, 2.0 ** (emin - mant - 2))))
    ex = np.maximum(ex, emin)                       # subnormals share the minimum exponent
    step = 2.0 ** (ex - mant)
    q = np.round(x / step) * step
    q = np.clip(q, -vmax, vmax)
    out = (q * s).reshape(shp[:-1] + (-1,))[..., :k]
    return np.moveaxis(out, -1, axis)


MODE = ['x3']


def mm(a, w):
    a = np.asarray(a, np.float32).astype(np.float64)      # activations are fp32 on the GPU
    w = np.asarray(w, np.float32).astype(np.float64)
    m = MODE[0]
    if m == 'exact':
        return a @ w
    ah, wh = f16(a), f16(w)
    al, wl = a - ah, w - wh
    if m == 'f16x3':
        return ah @ wh + ah @ f16(wl) + f16(al) @ wh
    if m == 'hi_only':
        return ah @ wh
    fmt = {'e4m3': dict(mant=3, emin=-6, vmax=448.0), 'e5m2': dict(mant=2, emin=-14, vmax=57344.0),
           'e2m3': dict(mant=3, emin=0, vmax=7.5), 'e3m2': dict(mant=2, emin=-2, vmax=28.0), 'e2m1': dict(mant=1, emin=0, vmax=6.0)}[m.split()[0]]
    A, W = a.reshape(-1, a.shape[-1]), w
    ah2, al2 = ah.reshape(A.shape), al.reshape(A.shape)
    y = ah2 @ wh + q8_blocks(ah2, 1, **fmt) @ q8_blocks(wl, 0, **fmt) + q8_blocks(al2, 1, **fmt) @ q8_blocks(wh, 0, **fmt)
    return y.reshape(a.shape[:-1] + (w.shape[1],))


def conv_emu(x, kernel, bias):
    kernel = np.asarray(kernel, np.float64)
    k = kernel.shape[1]
    b, l, c = x.shape
    lo = l - k + 1
    y = np.zeros((b, lo, kernel.shape[3]))
    for j in range(k):
        y += mm(x[:, j:j + lo, :], kernel[0, j])
    return y + np.asarray(bias, np.float64)


def dense_emu(x, kernel, bias):
    return mm(x, kernel) + np.asarray(bias, np.float64)


params = dict(synth.TDNN_STAT_PARAMS)
weights = synth.synth_weights(params, 30, seed=0)
feats = np.stack(synth.synth_features(4, 300, 30, seed=1234))
_, ep = ref_numpy.entire_network(feats, weights, params)
ref = ep["tdnn6_dense"]
ref_numpy.conv_valid, ref_numpy.dense_layer = conv_emu, dense_emu
for mode in ('exact', 'f16x3', 'hi_only', 'e4m3 cross terms', 'e5m2 cross terms', 'e2m3 cross terms (fp6)', 'e3m2 cross terms (fp6)', 'e2m1 cross terms (fp4)'):
    MODE[0] = mode
    _, e = ref_numpy.entire_network(feats, weights, params)
    err = np.linalg.norm(e["tdnn6_dense"] - ref, axis=1) / np.linalg.norm(ref, axis=1)
    print("%-24s tdnn6_dense rel-L2 max %.3e  mean %.3e" % (mode, err.max(), err.mean()))
