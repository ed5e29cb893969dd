#!/bin/bash
# Same-lease attribution of round 5's layout changes on the headline step (bench.py --no-cpu --no-epoch --no-v2 --no-alt, 40 steps), alternating:
#   caller   PYGAT_RENUMBER=0                 every node table in the caller's order (round 4's layout)
#   order    PYGAT_TAIL=0                     internal degree order, the self-loop-only rows through the fused kernels
#   tail     (default)                        internal degree order + tail streams (K3a on the rows before the tail)
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab_node_order}; mkdir -p $O; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in caller order tail; do
    case $v in caller) E="PYGAT_RENUMBER=0";; order) E="PYGAT_TAIL=0";; tail) E="PYGAT_NOOP=1";; esac
    env $E python3 bench.py --no-cpu --no-epoch --no-v2 --no-alt --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', round(d['ms_per_step'],4), {k['kernel']: round(k['avg_ms'],4) for k in d['kernels']})"
  done
done
