O=$GRAFT_REPO_ROOT/gpurun_out/r3j; mkdir -p $O; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_integration_stub.py tests/test_gpu_graphed.py tests/test_gpu_dropout.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" > $O/rc.txt
for n in 8 4 2; do python3 bench.py --as-rank-of $n --no-cpu --no-epoch --steps 20 > $O/r$n.json 2> $O/r$n.err; done
python3 bench.py --no-cpu --steps 20 > $O/bench.json 2> $O/bench.err
python3 tools/kbench.py --graph pubmed > $O/kbench_pubmed.log 2>&1
python3 tools/kbench.py --graph cora > $O/kbench_cora.log 2>&1
