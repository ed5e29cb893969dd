O=$GRAFT_REPO_ROOT/gpurun_out/r3k; mkdir -p $O; cd $GRAFT_REPO_ROOT
for bt in 64 128 256; do for n in 8 4; do PYGAT_NARROW_BLOCK=$bt python3 bench.py --as-rank-of $n --no-cpu --no-epoch --steps 20 > $O/r${n}_bt$bt.json 2> $O/r${n}_bt$bt.err; done; done
