#!/usr/bin/env python3
"""bench.py with module-level switches of pygat_amd.ops set first (A/B of Python-side experiments on one box):
    python3 tools/run_bench_with.py TAIL_OVERLAP=True RENUMBER_MIN_BYTES=0 -- --no-cpu --no-epoch --no-v2"""
import ast
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygat_amd.ops as ops  # noqa: E402

args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
for kv in args[:cut]:
    k, v = kv.split("=", 1)
    assert hasattr(ops, k), k
    setattr(ops, k, ast.literal_eval(v))
sys.argv = ["bench.py"] + args[cut + 1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
