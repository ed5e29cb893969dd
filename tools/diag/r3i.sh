O=$GRAFT_REPO_ROOT/gpurun_out/r3i; mkdir -p $O; cd $GRAFT_REPO_ROOT
for ts in 8 16 32 64; do PYGAT_SLOT_EDGES=$ts python3 bench.py --as-rank-of 8 --no-cpu --no-epoch --steps 20 > $O/r8_ts$ts.json 2> $O/r8_ts$ts.err; done
for ts in 16 32 64; do PYGAT_SLOT_EDGES=$ts python3 bench.py --as-rank-of 4 --no-cpu --no-epoch --steps 20 > $O/r4_ts$ts.json 2> $O/r4_ts$ts.err; done
