O=$GRAFT_REPO_ROOT/gpurun_out/r3t; mkdir -p $O; cd $GRAFT_REPO_ROOT
for rep in 1 2; do for nm in 0 1; do PYGAT_NO_SLOT_META=$nm python3 bench.py --no-cpu --no-epoch --no-v2 --steps 20 > $O/meta${nm}_$rep.json 2> $O/e.err; done; done
