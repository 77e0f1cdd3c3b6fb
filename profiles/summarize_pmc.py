#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per (kernel, grid).
usage: summarize_pmc.py <counter_collection.csv> [name-substring]"""
import csv
import sys
from collections import OrderedDict, defaultdict


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = OrderedDict()
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if flt and flt not in name:
                continue
            key = (name, row["Grid_Size"] if "Grid_Size" in row else row.get("Grid_Size_X", ""))
            d = acc.setdefault(key, defaultdict(list))
            d[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for key, d in acc.items():
        print("%s grid=%s" % key)
        for c, v in d.items():
            print("    %-28s n=%-3d mean=%.4g" % (c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
