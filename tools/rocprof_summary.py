#!/usr/bin/env python
"""Summarise rocprofv3 CSV output (kernel trace and/or counter collection) per kernel.

    python tools/rocprof_summary.py <dir> [<dir> ...]  > profiles/rNN_xxx.txt

Prints, per kernel name: calls, average / min / max duration (kernel-trace CSVs) and the mean of
every collected counter per dispatch (counter-collection CSVs).  FETCH_SIZE / WRITE_SIZE are in KiB
as rocprofv3 reports them; the gfx950 correction (FETCH_SIZE counts 64 B per 128-B request for wide
coalesced reads, MI355X_MICROARCH.md "HBM") is applied by the caller, not here.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    pre = ""
    if name.startswith("[grid "):
        pre, name = name.split("] ", 1)
        pre += "] "
    name = name.split("(")[0]
    return pre + name[-70:]


def main(dirs):
    by_grid = "--by-grid" in dirs          # one row per (kernel, grid size): the launches of different benchmark entries apart
    dirs = [d for d in dirs if d != "--by-grid"]
    for d in dirs:
        print("== {}".format(d))
        for path in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
            dur = defaultdict(list)
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row["Kernel_Name"]
                    if by_grid and "Grid_Size_X" in row:
                        name = "[grid {}x{}] ".format(row["Grid_Size_X"], row.get("Grid_Size_Y", "1")) + name
                    dur[name].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            print("-- kernel trace: {}".format(os.path.relpath(path, d)))
            print("{:>8} {:>12} {:>12} {:>12} {:>14}  {}".format("calls", "avg_us", "min_us", "max_us", "total_ms", "kernel"))
            for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
                print("{:>8} {:>12.3f} {:>12.3f} {:>12.3f} {:>14.3f}  {}".format(
                    len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, sum(v) / 1e6, short(k)))
        for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            acc = defaultdict(lambda: defaultdict(list))
            with open(path) as f:
                for row in csv.DictReader(f):
                    acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
            print("-- counters: {}".format(os.path.relpath(path, d)))
            for k, cs in acc.items():
                for c, v in sorted(cs.items()):
                    print("{:>8} {:>20.1f}  {:<28} {}".format(len(v), sum(v) / len(v), c, short(k)))


if __name__ == "__main__":
    main(sys.argv[1:])
