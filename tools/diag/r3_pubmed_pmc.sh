#!/bin/bash
# BASELINE config 3: HBM traffic of the Pubmed epoch's kernels (rocprofv3 --pmc, separate passes) + their durations.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3pm; P=$O/out; mkdir -p $O $P; cd $R; export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/epoch_profile.py pubmed --epochs 20 > $O/fetch.log 2>&1)
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/write -- python3 $R/tools/epoch_profile.py pubmed --epochs 20 > $O/write.log 2>&1)
python3 tools/pmc_summary.py $O/fetch $O/write --out $P/r3end_pubmed_epoch_pmc.json > $O/pmc_summary.log 2>&1; tail -3 $O/pmc_summary.log
rm -rf $O/fetch $O/write
echo done
