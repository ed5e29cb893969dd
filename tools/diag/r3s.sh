O=$GRAFT_REPO_ROOT/gpurun_out/r3s; mkdir -p $O; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_sparse_features.py tests/test_gpu_parity.py tests/test_gpu_dropout.py tests/test_gpu_training.py tests/test_gpu_graphed.py tests/test_gpu_ppi.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" > $O/rc.txt
python3 bench.py --no-cpu --no-v2 --steps 5 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/cora_prof -- python3 $GRAFT_REPO_ROOT/tools/epoch_profile.py cora --epochs 30 > $O/cora_prof.log 2>&1
cd $GRAFT_REPO_ROOT; python3 tools/epoch_sequence.py $O/cora_prof --epochs 35 > $O/cora_seq.txt 2>&1; rm -rf $O/cora_prof
