#!/bin/bash
# Evidence pass of a round: bench line, rocprofv3 kernel stats + PMC traffic of the same command, per-rank shard table, F' sweep,
# epoch kernel profiles, GEMM bench.  Run on the GPU box from the repo root:  bash tools/profile_round.sh <tag> [part]
# Writes under gpurun_out/<tag>/ and copies the summaries to profiles/<tag>_*.
TAG=$1; PART=${2:-all}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; P=$O/out   # (only gpurun_out/ travels back: copy $P/* to profiles/ afterwards)
mkdir -p $O $P; cd $R
export TMPDIR=/tmp
prof() { (cd /tmp && timeout -k 10 300 rocprofv3 "$@"); }
if [ "$PART" = all ] || [ "$PART" = a ]; then
  python3 bench.py > $O/bench.json 2> $O/bench.err; cp $O/bench.json $P/${TAG}_bench.json
  prof --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu --no-epoch --no-v2 --no-alt > $O/bench_under_rocprof.json 2> $O/stats.err
  python3 tools/rocprof_top.py $O/stats --top 14 --csv $P/${TAG}_kernel_stats.csv > $O/kernel_top.txt 2>&1; cp $O/bench_under_rocprof.json $P/${TAG}_bench_under_rocprof.json; rm -rf $O/stats
  prof --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --no-cpu --no-epoch --no-v2 --no-alt --steps 5 --warmup 2 > $O/pmc_fetch.log 2>&1
  prof --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/write -- python3 $R/bench.py --no-cpu --no-epoch --no-v2 --no-alt --steps 5 --warmup 2 > $O/pmc_write.log 2>&1
  python3 tools/pmc_summary.py $O/fetch $O/write --out $P/${TAG}_pmc_bench.json > $O/pmc_summary.log 2>&1; rm -rf $O/fetch $O/write
  prof --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq1 -- python3 $R/bench.py --no-cpu --no-epoch --no-v2 --no-alt --steps 4 --warmup 2 > $O/sq1.log 2>&1
  prof --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq2 -- python3 $R/bench.py --no-cpu --no-epoch --no-v2 --no-alt --steps 4 --warmup 2 > $O/sq2.log 2>&1
  python3 tools/sq_summary.py $O/sq1 $O/sq2 --out $P/${TAG}_sq_counters.json > $O/sq_summary.log 2>&1; rm -rf $O/sq1 $O/sq2
  for n in 8 4 2; do python3 bench.py --as-rank-of $n --no-cpu --no-epoch --no-v2 --steps 20 2> $O/rank$n.err; done > $P/${TAG}_as_rank_of.jsonl
  # the shard of an 8-GPU run at wider heads: rank-0 work against the 1-GPU step of the same F'
  for f in 64 128; do
    python3 bench.py --fout $f --as-rank-of 8 --no-cpu --no-epoch --no-v2 --steps 10 2> $O/rank8_f$f.err
  done > $P/${TAG}_as_rank_of_8_wide_heads.jsonl
  python3 bench.py --dx --no-cpu --no-epoch --steps 20 > $P/${TAG}_bench_dx.json 2> $O/dx.err
fi
if [ "$PART" = all ] || [ "$PART" = b ]; then
  for f in 8 64 128; do python3 bench.py --fout $f --no-cpu --no-epoch --no-v2 --steps 10 2> $O/fout$f.err; done > $P/${TAG}_fout_sweep.jsonl
  python3 tools/gemm_bench.py > $P/${TAG}_gemm_bench.log 2> $O/gemm.err
  for c in ppi cora pubmed; do
    prof --kernel-trace --stats --output-format csv -d $O/${c}_prof -- python3 $R/tools/epoch_profile.py $c --epochs 30 > $O/${c}_prof.log 2>&1
    python3 tools/rocprof_top.py $O/${c}_prof --per 35 --top 30 --csv $P/${TAG}_${c}_epoch_kernel_stats.csv > $P/${TAG}_${c}_epoch_top_kernels.txt 2>&1
    python3 tools/epoch_sequence.py $O/${c}_prof --epochs 35 > $P/${TAG}_${c}_epoch_sequence.txt 2>&1; rm -rf $O/${c}_prof
  done
  prof --kernel-trace --stats --output-format csv -d $O/v2 -- python3 $R/tools/v2_bench.py > $O/v2.log 2>&1
  python3 tools/rocprof_top.py $O/v2 --per 25 --top 10 --csv $P/${TAG}_gatv2_kernel_stats.csv > $P/${TAG}_gatv2_top_kernels.txt 2>&1; rm -rf $O/v2; grep ms_per_step $O/v2.log >> $P/${TAG}_gatv2_top_kernels.txt
  # (round 5: `bench.py --gpus N` starts its own ranks -- no launcher)
  BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --warmup 2 --verify > $P/${TAG}_gloo_rehearsal_2ranks.json 2> $P/${TAG}_gloo_rehearsal_2ranks.log
fi
if [ "$PART" = all ] || [ "$PART" = c ]; then
  # the per-rank shapes of the driver's 8-GPU default (ONE 16-float head per rank: 32-edge slots, narrow-row kernels) at the most
  # ranks one card may host (the pool's process guard admits 6 processes with the GPU open, the launcher included -- a 6-rank
  # run was killed by it): 5 heads on 5 ranks, and 4 on 4
  for n in 5 4; do
    BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python3 bench.py --gpus $n --heads $n --steps 3 --warmup 1 --verify > $P/${TAG}_gloo_rehearsal_${n}ranks_one_head_each.json 2> $P/${TAG}_gloo_rehearsal_${n}ranks_one_head_each.log
  done
  # config 4: PMC traffic of the PPI epoch's kernels (separate passes, as for configs 3 and 5)
  prof --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/ppi_fetch -- python3 $R/tools/epoch_profile.py ppi --epochs 12 > $O/ppi_fetch.log 2>&1
  prof --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/ppi_write -- python3 $R/tools/epoch_profile.py ppi --epochs 12 > $O/ppi_write.log 2>&1
  python3 tools/pmc_summary.py $O/ppi_fetch $O/ppi_write --out $P/${TAG}_ppi_epoch_pmc.json --edges 0 --heads 4 --note "PPI-shaped epoch (BASELINE config 4), tools/epoch_profile.py ppi: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE+TCC in separate passes, per-launch means, FETCH_SIZE doubled" > $O/ppi_pmc.log 2>&1; rm -rf $O/ppi_fetch $O/ppi_write
fi
if [ "$PART" = all ] || [ "$PART" = p ]; then
  # PMC traffic of the headline command alone (separate passes, no other tracing), and the promoted replay file
  prof --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --no-cpu --no-epoch --no-v2 --no-alt --steps 5 --warmup 2 > $O/pmc_fetch.log 2>&1
  prof --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/write -- python3 $R/bench.py --no-cpu --no-epoch --no-v2 --no-alt --steps 5 --warmup 2 > $O/pmc_write.log 2>&1
  python3 tools/pmc_summary.py $O/fetch $O/write --out $P/${TAG}_pmc_bench.json > $O/pmc_summary.log 2>&1; rm -rf $O/fetch $O/write
fi
if [ "$PART" = all ] || [ "$PART" = d ]; then
  # the N > 1 step through RCCL at the one world size a 1-GPU box hosts (BENCH_FORCE_DIST=1: the copy-free exchange path, chunked K2,
  # RCCL's in-place all-gather of each chunk view), at 1 / 2 / 4 row chunks, and its kernel trace: no copy / permute kernel in it
  for c in 1 2 4; do BENCH_FORCE_DIST=1 python3 bench.py --no-cpu --no-epoch --no-v2 --chunks $c 2> $O/rccl1_$c.err; done > $P/${TAG}_bench_rccl_world1_chunks_1_2_4.jsonl
  BENCH_FORCE_DIST=1 prof --kernel-trace --stats --output-format csv -d $O/rccl1 -- python3 $R/bench.py --no-cpu --no-epoch --no-v2 > $O/rccl1_under_rocprof.json 2> $O/rccl1_stats.err
  python3 tools/rocprof_top.py $O/rccl1 --top 40 > $P/${TAG}_rccl_world1_kernels.txt 2>&1; rm -rf $O/rccl1
fi
echo "profile_round $TAG $PART done"
