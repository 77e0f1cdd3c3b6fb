#!/usr/bin/env python
"""Where the device sits idle under the command-line driver: rocprofv3 --kernel-trace of one extract.py run, then the gaps between
consecutive kernels on the timeline.  Two steps (the profiled program must be the driver itself, not a launcher of it):
  python tools/cli_trace_gaps.py make <dir> [n_utts]          -> <dir>/exp (model), <dir>/feats.ark
  rocprofv3 --kernel-trace --output-format csv -d <out> -- python -m tf_kaldi_speaker_amd.extract --gpu 0 <dir>/exp ark:<dir>/feats.ark ark:<dir>/x.ark
  python tools/cli_trace_gaps.py gaps <kernel_trace.csv>"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make(d, n):
    import numpy as np
    from tf_kaldi_speaker_amd import kaldi_io, model_io, synth
    os.makedirs(d, exist_ok=True)
    params = dict(synth.TDNN_STAT_PARAMS)
    model_io.save_model(os.path.join(d, "exp"), params, 30, synth.synth_weights(params, 30, seed=0), step=1)
    rs = np.random.RandomState(0)
    base = rs.standard_normal((364, 30)).astype(np.float32)
    with open(os.path.join(d, "feats.ark"), "wb") as f:
        for i in range(n):
            kaldi_io.write_mat(f, base[i % 64:i % 64 + 300], key="utt%07d" % i)


def gaps(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the steady part: from the first to the last two-unit 7-tap launch
    idx = [i for i, r in enumerate(rows) if "gemm_f6v2_kernel<7" in r[2]]
    rows = rows[idx[2]:idx[-1] + 1]
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    g, end = [], rows[0][1]
    for s, e, name in rows[1:]:
        if s > end:
            g.append((s - end, name))
        end = max(end, e)
    idle = sum(x for x, _ in g)
    print("kernels %d  span %.3f s  sum of kernel times %.3f s  idle between kernels %.3f s (%.1f %%)"
          % (len(rows), span / 1e9, busy / 1e9, idle / 1e9, 100.0 * idle / span))
    by = {}
    for x, name in g:
        k = name.split("(")[0][:60]
        by.setdefault(k, [0, 0])
        by[k][0] += x
        by[k][1] += 1
    for k, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:8]:
        print("  idle before %-60s %8.3f ms in %5d gaps (%.1f us each)" % (k, t / 1e6, c, t / c / 1e3))
    per = {}
    for s, e, name in rows:
        k = name.split("(")[0][:60]
        per.setdefault(k, [0, 0])
        per[k][0] += e - s
        per[k][1] += 1
    for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:12]:
        print("  %-60s %8.3f ms  %6d calls  %.1f us" % (k, t / 1e6, c, t / c / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "make":
        make(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 100000)
    else:
        gaps(sys.argv[2])
