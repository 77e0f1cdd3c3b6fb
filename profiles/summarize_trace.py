#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid): the five TDNN layers share one
GEMM kernel symbol, so `--stats` alone lumps them; the grid size tells the launches apart.  When two layers of a
forward pass also share the grid (since the tail K-split, L2 and L3 both launch 2304 main tiles), the group holds
k launches per forward and is split into k sub-rows `#0 .. #k-1` by launch order (forward-pass order).
usage: summarize_trace.py <kernel_trace.csv> [skip_first_n_per_group]"""
import csv
import sys
from collections import OrderedDict


def split_by_order(groups, nfwd):
    """{key: [records in launch order]} -> same with keys (key, sub) where a group repeats k times per forward."""
    out = OrderedDict()
    for key, recs in groups.items():
        k = len(recs) // nfwd if nfwd and len(recs) % nfwd == 0 else 1
        if k <= 1 or "gemm" not in key[0]:
            out[(key, None)] = recs
            continue
        for j in range(k):
            out[(key, j)] = recs[j::k]
    return out


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = []
    with open(path) as f:
        for row in csv.DictReader(f):
            rows.append(row)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    groups = OrderedDict()
    for row in rows:
        short = row["Kernel_Name"].split("(")[0].replace("void ", "")
        key = (short, int(row["Grid_Size_X"]), int(row["Grid_Size_Y"]), int(row["Grid_Size_Z"]), int(row["Workgroup_Size_X"]))
        groups.setdefault(key, []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"]),
                                           row["VGPR_Count"], row["LDS_Block_Size"]))
    gemm_calls = [len(v) for k, v in groups.items() if "gemm" in k[0]]
    nfwd = min(gemm_calls) if gemm_calls else 0
    print("%-44s %-18s %6s %12s %12s %12s  %5s %7s" % ("kernel", "grid(threads)", "calls", "avg_us", "min_us", "max_us", "vgpr", "lds"))
    for (key, sub), recs in split_by_order(groups, nfwd).items():
        d = [r[0] for r in recs][skip:] or [r[0] for r in recs]
        name = key[0][:40] + ("" if sub is None else " #%d" % sub)
        print("%-44s %-18s %6d %12.1f %12.1f %12.1f  %5s %7s" % (
            name, "%dx%dx%d/%d" % key[1:], len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3,
            recs[0][1], recs[0][2]))


if __name__ == "__main__":
    main()
