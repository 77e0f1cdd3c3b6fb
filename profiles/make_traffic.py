#!/usr/bin/env python
"""Build traffic.json (HBM-side bytes per launch of the five TDNN layer GEMMs) from the FETCH_SIZE / WRITE_SIZE
summaries written by summarize_pmc.py.  usage: make_traffic.py <round dir> [precision of the profiled run: f16f6 | bf16x3]
Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md: both counters are in KB;
FETCH_SIZE under-reports 16 B/lane streaming reads by 2x on gfx950, so it is doubled.
A layer is the sum of its launches: tiles (whole tiles and the K-split slices of the tail tiles share one launch) + the tail reduce."""
import json
import re
import sys

# 256 x [300, 30]: (kernel substring, threads per launch, sub-index or None) -> layer
PARTS = {
    "bf16x3": [
        ("im2col_sb_kernel", 1228800, None, "tdnn1_conv"),                                           # feature rows -> SB rows (staging of L1)
        ("w14p2_kernel", 614400, None, "tdnn1_conv"),                                                # L1: 5 taps over 32-channel padded rows
        ("w14p2_kernel", 655360, 0, "tdnn2_conv"), ("w14p2_kernel", 655360, 1, "tdnn3_conv"),           # whole tiles + K-split tail slices
        ("tail_reduce_kernel", 262144, None, "tdnn2_conv"), ("tail_reduce_kernel", 131072, None, "tdnn3_conv"),
        ("w1p3_kernel", 585728, None, "tdnn4_dense"), ("w1p3_kernel", 1757184, None, "tdnn5_dense"),
    ],
    "f16f6": [
        ("im2col_sb_kernel", 1228800, None, "tdnn1_conv"),                                           # feature rows -> SB rows (staging of L1)
        ("w14p2_kernel", 614400, None, "tdnn1_conv"),                                                # L1 on the f16 kernel, F6-output epilogue
        ("gemm_f6v2_kernel<5", None, None, "tdnn2_conv"),                                            # whole tiles + K-split tail slices (any grid)
        ("gemm_f6v2_kernel<7", None, None, "tdnn3_conv"),
        ("f6v2_tail_reduce_kernel", 262144, None, "tdnn2_conv"), ("f6v2_tail_reduce_kernel", 131072, None, "tdnn3_conv"),
        ("w1p3_kernel", 585728, None, "tdnn4_dense"), ("w1p3_kernel", 1757184, None, "tdnn5_dense"),
    ],
}


def read(path, counter, parts):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"(\S.*?)(?: #(\d+))? grid=(\d+)", line)
        if m:
            cur = None
            for sub, grid, idx, layer in parts:
                if sub in m.group(1) and (grid is None or int(m.group(3)) == grid) and \
                        (idx is None or (m.group(2) is not None and int(m.group(2)) == idx)):
                    cur = layer
            continue
        m = re.match(r"\s+%s\s+n=\d+\s+mean=(\S+)" % counter, line)
        if m and cur:
            out[cur] = out.get(cur, 0.0) + float(m.group(1)) * 1024.0
    return out


def main():
    d = sys.argv[1]
    precision = sys.argv[2] if len(sys.argv) > 2 else "f16f6"      # precision of the profiled command (bench.py default: f16f6)
    fetch = read(d + "/pmc_fetch_summary.txt", "FETCH_SIZE", PARTS[precision])
    write = read(d + "/pmc_write_summary.txt", "WRITE_SIZE", PARTS[precision])
    kernels = {}
    for name in ("tdnn1_conv", "tdnn2_conv", "tdnn3_conv", "tdnn4_dense", "tdnn5_dense"):
        if name in fetch and name in write:
            kernels[name] = {"fetch_bytes": 2 * fetch[name], "write_bytes": write[name],
                             "traffic_bytes": 2 * fetch[name] + write[name]}
    json.dump({"note": "HBM-side bytes per layer launch (main tiles + tail K-split slices + reduce) from rocprofv3 --pmc "
                       "FETCH_SIZE / WRITE_SIZE (separate passes, KB units); FETCH_SIZE doubled per MI355X_MICROARCH.md "
                       "(16 B/lane streaming reads report 1/2 on gfx950)",
               "config": "%s, 256 x [300,30]" % precision, "kernels": kernels}, open(d + "/traffic.json", "w"), indent=1)
    for k, v in kernels.items():
        print("%-12s fetch %7.1f MB  write %7.1f MB  total %7.1f MB" % (k, v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6,
                                                                       v["traffic_bytes"] / 1e6))


if __name__ == "__main__":
    main()
