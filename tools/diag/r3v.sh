O=$GRAFT_REPO_ROOT/gpurun_out/r3v; mkdir -p $O; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_dropout.py tests/test_gpu_training.py tests/test_gpu_graphed.py tests/test_gpu_sparse_features.py tests/test_gpu_parity.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" > $O/rc.txt
python3 bench.py --no-cpu --no-v2 --steps 5 > $O/bench.json 2> $O/bench.err
