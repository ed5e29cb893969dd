set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $O
cd $GRAFT_REPO_ROOT
for n in 8 4 2; do python3 bench.py --as-rank-of $n --no-cpu --no-epoch --steps 20 > $O/as_rank_of_$n.json 2> $O/as_rank_of_$n.err; done
BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 2 --verify > $O/gloo2.json 2> $O/gloo2.err
echo "gloo2 rc $?" > $O/rc.txt
BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 4 --steps 3 --warmup 1 --verify > $O/gloo4.json 2> $O/gloo4.err
echo "gloo4 rc $?" >> $O/rc.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --as-rank-of 8 --no-cpu --no-epoch --steps 5 --warmup 2 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/write -- python3 $GRAFT_REPO_ROOT/bench.py --as-rank-of 8 --no-cpu --no-epoch --steps 5 --warmup 2 > $O/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $O/fetch $O/write --out $O/pmc_rank8.json --heads 1 > $O/pmc_summary.log 2>&1
rm -rf $O/fetch $O/write
cat $O/rc.txt
