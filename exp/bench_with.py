#!/usr/bin/env python3
"""bench.py with module attributes of svox_t_amd.csrc set first (switches that are not environment variables):
   python exp/bench_with.py MASK_CLEARS_LISTS=False -- --no-cpu-baseline --no-plain"""
import ast, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import svox_t_amd.csrc as C
i = sys.argv.index("--") if "--" in sys.argv else len(sys.argv)
for kv in sys.argv[1:i]:
    k, v = kv.split("=", 1)
    assert hasattr(C, k), k
    setattr(C, k, ast.literal_eval(v))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[i + 1:]
runpy.run_path(sys.argv[0], run_name="__main__")
