O=$GRAFT_REPO_ROOT/gpurun_out/r3u; mkdir -p $O; cd $GRAFT_REPO_ROOT
for ts in 8 16 32 64; do PYGAT_SLOT_EDGES=$ts python3 tools/epoch_profile.py ppi --spans > $O/ppi_ts$ts.log 2>&1; done
