#!/usr/bin/env python
"""Build traffic.json (HBM-side bytes per launch of the five TDNN layer GEMMs) from the FETCH_SIZE / WRITE_SIZE
summaries written by summarize_pmc.py.  usage: make_traffic.py <round dir>
Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md: both counters are in KB;
FETCH_SIZE under-reports 16 B/lane streaming reads by 2x on gfx950, so it is doubled."""
import json
import re
import sys

LAYER_BY_GRID = {614400: "tdnn1_conv", 606208: "tdnn2_conv", 598016: "tdnn3_conv", 585728: "tdnn4_dense",
                 1757184: "tdnn5_dense"}          # 256 x [300, 30], threads per launch


def read(path, counter):
    out, grid = {}, None
    for line in open(path):
        m = re.match(r"\S.*grid=(\d+)", line)
        if m:
            grid = int(m.group(1)) if "gemm_bf16x3" in line else None
            continue
        m = re.match(r"\s+%s\s+n=\d+\s+mean=(\S+)" % counter, line)
        if m and grid in LAYER_BY_GRID:
            out[LAYER_BY_GRID[grid]] = float(m.group(1)) * 1024.0
    return out


def main():
    d = sys.argv[1]
    fetch = read(d + "/pmc_fetch_summary.txt", "FETCH_SIZE")
    write = read(d + "/pmc_write_summary.txt", "WRITE_SIZE")
    kernels = {}
    for name in LAYER_BY_GRID.values():
        if name in fetch and name in write:
            kernels[name] = {"fetch_bytes": 2 * fetch[name], "write_bytes": write[name],
                             "traffic_bytes": 2 * fetch[name] + write[name]}
    json.dump({"note": "HBM-side bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB units); "
                       "FETCH_SIZE doubled per MI355X_MICROARCH.md (16 B/lane streaming reads report 1/2 on gfx950)",
               "config": "bf16x3, 256 x [300,30]", "kernels": kernels}, open(d + "/traffic.json", "w"), indent=1)
    for k, v in kernels.items():
        print("%-12s fetch %7.1f MB  write %7.1f MB  total %7.1f MB" % (k, v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6,
                                                                       v["traffic_bytes"] / 1e6))


if __name__ == "__main__":
    main()
