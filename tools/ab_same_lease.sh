#!/bin/bash
# Same-lease A/B of two builds of the headline step: runs bench.py of tree A (this repo) and of
# tree B (a copy of an older commit staged under _ab_r2/, built in the container) alternately on ONE box, so that the
# pool's box-to-box spread cannot pose as a kernel change.   bash tools/ab_same_lease.sh <tag> [rounds]
TAG=${1:-ab}; ROUNDS=${2:-3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export TMPDIR=/tmp
for i in $(seq $ROUNDS); do
  (cd $R/_ab_r2 && python3 bench.py --no-cpu --no-epoch --steps 40 --warmup 10) >> $O/B.jsonl 2>> $O/B.err || exit 1
  (cd $R && python3 bench.py --no-cpu --no-epoch --no-v2 --steps 40 --warmup 10) >> $O/A.jsonl 2>> $O/A.err || exit 1
  echo "round $i done"
done
# kernel-trace averages of both under the profiler
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/statsB -- python3 $R/_ab_r2/bench.py --no-cpu --no-epoch --steps 20) > $O/B_rocprof.json 2> $O/B_rocprof.err
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/statsA -- python3 $R/bench.py --no-cpu --no-epoch --no-v2 --steps 20) > $O/A_rocprof.json 2> $O/A_rocprof.err
python3 tools/rocprof_top.py $O/statsB --top 10 > $O/B_top.txt 2>&1
python3 tools/rocprof_top.py $O/statsA --top 10 > $O/A_top.txt 2>&1
rm -rf $O/statsA $O/statsB
python3 - <<EOF
import json
for t in "AB":
    for l in open("$O/%s.jsonl" % t):
        d = json.loads(l)
        ks = {k["kernel"]: round(k["avg_ms"], 4) for k in d.get("kernels", [])}
        print(t, d["ms_per_step"], ks)
EOF
