#!/usr/bin/env python
"""Per-pattern summary of tools/prof_bench.sh: one row per sampling pattern for the headline kernel -- calls, avg / min / max
duration from the rocprofv3 kernel trace, the fraction of the 8 TB/s roofline that average corresponds to (27 B/px
algorithmic), the `roofline.frac` bench.py printed in the same (profiled) run, and the HBM bytes per field from the FETCH_SIZE
and WRITE_SIZE passes (2 x FETCH + WRITE: gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md "HBM").  Also
writes <dir>/traffic.json, which bench.py reads as `roofline.traffic` once it is committed under profiles/.

    python tools/prof_bench_summary.py gpurun_out/prof_<tag>  > profiles/rNN_compose3_rocprof_summary.txt
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNEL = "compose3_xpose_kernel"


def durations(d):
    dur = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                dur[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    return dur


def counters(d):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def main(out):
    traffic = {"source": "tools/prof_bench.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, one sampling pattern per run; "
                         "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), gfx950 correction of MI355X_MICROARCH.md", "patterns": {}}
    print("{:<7} {:>6} {:>10} {:>10} {:>10} {:>10} {:>11} {:>13} {:>13} {:>14} {:>9}".format(
        "pattern", "calls", "avg_us", "min_us", "max_us", "frac(avg)", "frac(line)", "FETCH_KiB", "WRITE_KiB", "HBM_MB/field", "x algo"))
    for pat in ("scale", "shift", "rot"):
        d = os.path.join(out, pat)
        line = {}
        try:
            line = json.loads(open(os.path.join(out, pat + ".bench_line.json")).read())
        except Exception:      # noqa: BLE001
            pass
        algo = line.get("roofline", {}).get("algorithmic_bytes_per_launch", 27 * 2160 * 3840 * 8)
        batch = line.get("config", {}).get("fields_per_step_per_gpu", 8)
        dur = [v for k, v in durations(os.path.join(d, "trace")).items() if KERNEL in k]
        dur = dur[0] if dur else []
        cs = {}
        for k, v in counters(d).items():
            if KERNEL in k:
                for c, vals in v.items():
                    cs[c] = sum(vals) / len(vals)
        avg = sum(dur) / len(dur) / 1e3 if dur else float("nan")
        fetch, write = cs.get("FETCH_SIZE", float("nan")), cs.get("WRITE_SIZE", float("nan"))
        hbm = (2 * fetch + write) * 1024 / batch
        print("{:<7} {:>6} {:>10.3f} {:>10.3f} {:>10.3f} {:>10.4f} {:>11} {:>13.1f} {:>13.1f} {:>14.2f} {:>9.3f}".format(
            pat, len(dur), avg, min(dur) / 1e3 if dur else float("nan"), max(dur) / 1e3 if dur else float("nan"),
            algo / (avg * 1e-6) / 8e12 if dur else float("nan"), line.get("roofline", {}).get("frac", "-"),
            fetch, write, hbm / 1e6, hbm / (algo / batch)))
        traffic["patterns"][pat] = {"hbm_bytes_per_field": round(hbm), "fetch_kib_per_launch": round(fetch, 1), "write_kib_per_launch": round(write, 1),
                                    "fields_per_launch": batch, "ratio_to_algorithmic": round(hbm / (algo / batch), 4),
                                    "trace_avg_us": round(avg, 3), "trace_calls": len(dur)}
    if "scale" in traffic["patterns"]:
        traffic["hbm_bytes_per_field"] = traffic["patterns"]["scale"]["hbm_bytes_per_field"]
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print("\nfrac(avg) = 27 B/px x px per launch / avg_us / 8 TB/s from THIS trace; frac(line) = what bench.py printed in the same profiled run "
          "(HIP events around the timed steps only; the trace also holds the warm-up launches).")
    print("\n== other kernels and the instruction / cache counters of the headline pattern (scale)")
    for k, v in sorted(durations(os.path.join(out, "scale", "trace")).items(), key=lambda kv: -sum(kv[1])):
        print("{:>8} calls {:>12.3f} us avg  {}".format(len(v), sum(v) / len(v) / 1e3, k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-80:]))
    for k, v in counters(os.path.join(out, "scale")).items():
        if KERNEL in k:
            for c, vals in sorted(v.items()):
                print("{:>8} {:>20.1f}  {}".format(len(vals), sum(vals) / len(vals), c))


if __name__ == "__main__":
    main(sys.argv[1])
