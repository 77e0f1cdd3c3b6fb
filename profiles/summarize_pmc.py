#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per (kernel, grid).  Groups that repeat k
times per forward pass (two layers with the same kernel and grid) are split into `#0 .. #k-1` by dispatch order.
usage: summarize_pmc.py <counter_collection.csv> [name-substring]"""
import csv
import sys
from collections import OrderedDict


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = []
    with open(path) as f:
        for row in csv.DictReader(f):
            rows.append(row)
    if rows and "Dispatch_Id" in rows[0]:
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    acc = OrderedDict()           # (name, grid) -> counter -> [values in dispatch order]
    for row in rows:
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if flt and flt not in name:
            continue
        key = (name, row["Grid_Size"] if "Grid_Size" in row else row.get("Grid_Size_X", ""))
        acc.setdefault(key, OrderedDict()).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    counts = [len(next(iter(d.values()))) for k, d in acc.items() if "gemm" in k[0]]
    nfwd = min(counts) if counts else 0
    for key, d in acc.items():
        n = len(next(iter(d.values())))
        k = n // nfwd if nfwd and n % nfwd == 0 and "gemm" in key[0] else 1
        for j in range(k):
            print("%s%s grid=%s" % (key[0], "" if k == 1 else " #%d" % j, key[1]))
            for c, v in d.items():
                vv = v[j::k]
                print("    %-28s n=%-3d mean=%.4g" % (c, len(vv), sum(vv) / len(vv)))


if __name__ == "__main__":
    main()
