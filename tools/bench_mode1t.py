#!/usr/bin/env python
"""combine_with(mode=1), ref 't', at 2160 x 3840 in a loop (for profiling): python tools/bench_mode1t.py [--iters N]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oflibnumpy_amd as of
from bench_ops import timed, report

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=6)
args = ap.parse_args()
of.native.ensure_device()
h, w = 2160, 3840
f2 = of.Flow.from_transforms([['scaling', 1000, 800, 0.9]], [h, w], 't').to_device()
f3 = of.Flow.from_transforms([['rotation', 1920, 1080, -20], ['scaling', 1000, 800, 0.9]], [h, w], 't').to_device()
f2.stats(); f3.stats()
t = timed(lambda: f2.combine_with(f3, 1), args.iters)
report("combine_with mode 1 't'", (h, w), 4 * 18 + 27 + 3 * 27, *t)
