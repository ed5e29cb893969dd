#!/bin/bash
# Same-lease A/B of variant builds of the library on the headline step (one GPU box, so that the pool's box-to-box spread cannot
# pose as a kernel change):
#     tools/build_variant.sh k4pf k4_backward_col.hip "-DPYGAT_K4_PREFETCH_ALL=1"        (in the build container)
#     gpurun -- 'bash tools/ab_variants.sh <tag> default k4pf [more variants ...]'          (on the GPU box)
# Runs bench.py (--no-cpu --no-epoch --no-v2, 40 steps) twice per variant, in the order given, and prints step time and the
# per-kernel HIP-event averages.  "default" = pygat_amd/libpygat_amd.so; any other name = pygat_amd/libpygat_amd_<name>.so.
# Extra bench flags: BENCH_FLAGS="--dx".  Results under gpurun_out/<tag>/bench_<variant>.jsonl.
TAG=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for v in "$@"; do
  lib=""; [ "$v" != default ] && lib=$R/pygat_amd/libpygat_amd_$v.so
  [ -z "$lib" ] || [ -f "$lib" ] || { echo "no such variant: $lib"; exit 1; }
  for i in 1 2; do
    PYGAT_AMD_LIB=$lib python3 bench.py --no-cpu --no-epoch --no-v2 --steps 40 --warmup 10 $BENCH_FLAGS >> $O/bench_$v.jsonl 2>> $O/bench.err || exit 1
  done
done
python3 - "$O" "$@" <<'PY'
import json, sys
O, names = sys.argv[1], sys.argv[2:]
for v in names:
    for l in open(f"{O}/bench_{v}.jsonl"):
        d = json.loads(l)
        print(f"{v:12s} {d['ms_per_step']:.4f} ms  " + "  ".join(f"{k['kernel']} {k['avg_ms']:.4f}" for k in d["kernels"]))
PY
