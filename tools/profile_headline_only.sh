TAG=r5z; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}c; P=$O/out; mkdir -p $O $P; cd $R; export TMPDIR=/tmp
prof() { (cd /tmp && timeout -k 10 300 rocprofv3 "$@"); }
prof --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu --no-epoch --no-v2 --no-alt > $O/bench_under_rocprof.json 2> $O/stats.err
python3 tools/rocprof_top.py $O/stats --top 16 --csv $P/${TAG}_kernel_stats.csv > $P/${TAG}_kernel_top.txt 2>&1; cp $O/bench_under_rocprof.json $P/${TAG}_bench_under_rocprof.json; rm -rf $O/stats
bash tools/profile_round.sh ${TAG}c p > $O/p.log 2>&1
cp $O/out/${TAG}c_pmc_bench.json $P/${TAG}_pmc_bench.json
cat $P/${TAG}_kernel_top.txt
