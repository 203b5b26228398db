"""A / B of two builds of the library on ONE box: the headline bench (10 steps)
with the in-tree libtce_hip.so (`cur`) or with scripts/variants/<name>.so (e.g. the
objects of the product build linked with one file of an earlier commit):
    python scripts/ab_lib.py cur | libtce_prev.so"""
import sys, os, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import _lib
if sys.argv[1].startswith("switch:"):
    # e.g. switch:tce_policy_inline_surrogate=0 -- a library switch before the run
    name, val = sys.argv[1][7:].split("=")
    getattr(_lib.load(), name)(int(val))
elif sys.argv[1] != "cur":
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", sys.argv[1])
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-configs", "--steps", "10", "--warmup", "4"]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
