R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4z2; P=$O/out; mkdir -p $P; cd $R
for n in 8 4 2; do python3 bench.py --as-rank-of $n --no-cpu --no-epoch --no-v2 --steps 20 2> $O/rank$n.err; done > $P/r4z_as_rank_of.jsonl
for f in 64 128; do python3 bench.py --fout $f --as-rank-of 8 --no-cpu --no-epoch --no-v2 --steps 10 2> $O/rank8_f$f.err; done > $P/r4z_as_rank_of_8_wide_heads.jsonl
python3 bench.py > $P/r4z_bench.json 2> $O/bench.err
python3 - <<PY
import json
for f in ("r4z_as_rank_of.jsonl","r4z_as_rank_of_8_wide_heads.jsonl"):
    for ln in open("$P/"+f):
        j=json.loads(ln); print(f, j["config"].get("f_out"), round(j["ms_per_step"],3), {k["kernel"]:round(k["avg_ms"],3) for k in j["kernels"]})
j=json.load(open("$P/r4z_bench.json")); print(j["ms_per_step"], {k["kernel"]:round(k["avg_ms"],4) for k in j["kernels"]}, j["epoch_ms"]["ppi"]["ms"], j["epoch_ms"]["cora"]["ms"])
PY
