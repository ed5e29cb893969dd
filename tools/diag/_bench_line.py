import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(sys.argv[1], round(d["ms_per_step"],3), [(k["kernel"][:8], round(k["avg_ms"],3)) for k in d["kernels"]])
