#!/usr/bin/env python
"""bench.py's cfg4 leg alone (run_cfg4), for A/Bs that need the leg's own figures without the other legs: from_x0, first iteration."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=10)
ap.add_argument("--iters", type=int, default=40)
a = ap.parse_args()
args = argparse.Namespace(cfg4_n=1e7)
leg = bench.run_cfg4(args, 0, 1, 0, None, None, iters=a.iters, m=a.m)
print(json.dumps({"value": leg["value"], "from_x0": leg["from_x0"]["value"], "first_iteration_ms": leg["from_x0"]["first_iteration_ms"]}))
