#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid): the five TDNN layers share one
GEMM kernel symbol, so `--stats` alone lumps them; the grid size tells the launches apart.
usage: summarize_trace.py <kernel_trace.csv> [skip_first_n_per_group]"""
import csv
import sys
from collections import OrderedDict


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    groups = OrderedDict()
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            short = name.split("(")[0].replace("void ", "")
            key = (short, int(row["Grid_Size_X"]), int(row["Grid_Size_Y"]), int(row["Grid_Size_Z"]),
                   int(row["Workgroup_Size_X"]))
            groups.setdefault(key, []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"]),
                                               row["VGPR_Count"], row["LDS_Block_Size"]))
    print("%-44s %-18s %6s %12s %12s %12s  %5s %7s" % ("kernel", "grid(threads)", "calls", "avg_us", "min_us", "max_us", "vgpr", "lds"))
    for key, recs in groups.items():
        d = [r[0] for r in recs][skip:] or [r[0] for r in recs]
        print("%-44s %-18s %6d %12.1f %12.1f %12.1f  %5s %7s" % (
            key[0][:44], "%dx%dx%d/%d" % key[1:], len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3,
            recs[0][1], recs[0][2]))


if __name__ == "__main__":
    main()
