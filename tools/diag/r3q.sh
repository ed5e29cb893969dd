O=$GRAFT_REPO_ROOT/gpurun_out/r3q; mkdir -p $O; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropout.py tests/test_gpu_training.py tests/test_gpu_graphed.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" > $O/rc.txt
for il in 1 0; do for ts in 16 32; do PYGAT_INLANE=$il PYGAT_SLOT_EDGES=$ts python3 bench.py --as-rank-of 8 --no-cpu --no-epoch --no-v2 --steps 20 > $O/r8_il${il}_ts$ts.json 2> $O/r8_il${il}_ts$ts.err; done; done
python3 bench.py --no-cpu --no-v2 --steps 5 > $O/bench.json 2> $O/bench.err
