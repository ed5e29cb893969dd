import sys, runpy, os
sys.path.insert(0, os.getcwd())
import pygat_amd.ops as o
o.RENUMBER_MIN_BYTES = 0
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
