"""Debug: per-workgroup phase timing of one traced bf16x3 GEMM launch (XVEC_TRACE_K=<K of the layer>).
Needs a library built with -DXV_GEMM_TRACE (XVEC_EXTRA_CFLAGS=-DXV_GEMM_TRACE python -c "import __graft_entry__ as g; g.build(force=True)").
usage: XVEC_TRACE_K=3584 [XVEC_TRACE_N=512] python tools/gemm_trace.py [self_attention]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import tf_kaldi_speaker_amd as xv  # noqa: E402
from tf_kaldi_speaker_amd import _lib, synth  # noqa: E402
from tf_kaldi_speaker_amd.params import Params  # noqa: E402
from tf_kaldi_speaker_amd.trainer import Trainer  # noqa: E402

params = Params(**dict(synth.TDNN_ATT_PARAMS if "self_attention" in sys.argv[1:] else synth.TDNN_STAT_PARAMS))
weights = synth.synth_weights(params, 30, seed=0)
tr = Trainer(params, None, 30, single_cpu=True, device=0, precision="bf16x3")
tr.build("predict")
tr.load_weights(weights)
B, T = 256, 300
feats = torch.randn(B * T, 30, device="cuda")
offs = np.arange(B + 1, dtype=np.int32) * T
for _ in range(5):
    tr.predict_packed(feats, offs)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((16384, 8), dtype=np.int64)
lib.xvdbg_gemm_trace.argtypes = [C.c_void_p, C.c_int]
n = lib.xvdbg_gemm_trace(buf.ctypes.data_as(C.c_void_p), 16384)
t = buf[:n].astype(np.float64) * 0.01          # 100 MHz -> microseconds
t0 = t[:, 0].min()
print("workgroups %d   kernel span %.1f us" % (n, t[:, 3].max() - t0))
for name, a, b in (("prologue", 0, 1), ("k-loop", 1, 2), ("epilogue", 2, 3), ("total", 0, 3)):
    d = t[:, b] - t[:, a]
    print("%-9s mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us" % (name, d.mean(), *np.percentile(d, [10, 50, 90]), d.max()))
# stamps inside the epilogue (wave 0): 4 parameters loaded, 5 split-blocked output done, 6 first fp32 / pooling pass staged,
# 7 first pooling pass done
sb_only = (t[:, 6] > 0).any() and (t[:, 6] <= t[:, 5]).all()       # stamps 6 / 7 inside the split-blocked pass 0
phases = ((("epi:params", 2, 4), ("epi:sb0 stage", 4, 6), ("epi:sb0 store", 6, 7), ("epi:sb1", 7, 5), ("epi:rest", 5, 3)) if sb_only else
          (("epi:params", 2, 4), ("epi:sb", 4, 5), ("epi:stage", 5, 6), ("epi:pool0", 6, 7), ("epi:rest", 7, 3)))
for name, a, b in phases:
    ok = (t[:, a] > 0) & (t[:, b] > 0)
    if ok.sum() > 0:
        d = (t[:, b] - t[:, a])[ok]
        print("%-14s mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f us  (%d wgs)" % (name, d.mean(), *np.percentile(d, [10, 50, 90]), ok.sum()))
start = np.sort(t[:, 0] - t0)
print("start times (us): p0 %.1f p25 %.1f p33 %.1f p34 %.1f p50 %.1f p66 %.1f p67 %.1f p75 %.1f p100 %.1f" %
      tuple(np.percentile(start, [0, 25, 33, 34, 50, 66, 67, 75, 100])))
hist, edges = np.histogram(t[:, 2] - t0, bins=24)
print("k-loop end histogram (us):", " ".join("%d@%.0f" % (h, e) for h, e in zip(hist, edges[:-1])))
