#!/bin/bash
# SQ / TA counter passes over one bench.py configuration (development tool; each rocprofv3 run collects its own few counters).
#   bash tools/diag/sq_passes.sh <out dir under gpurun_out> <bench args...>
O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $O/sq1 -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu --no-epoch --steps 4 --warmup 2 > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq2 -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu --no-epoch --steps 4 --warmup 2 > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $O/sq3 -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu --no-epoch --steps 4 --warmup 2 > $O/sq3.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/sq_summary.py $O/sq1 $O/sq2 $O/sq3 --out $O/sq.json > $O/sq_summary.log 2>&1
rm -rf $O/sq1 $O/sq2 $O/sq3
