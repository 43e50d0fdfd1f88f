#!/bin/bash
# rocprofv3 passes over the bench command (run on the GPU box through gpurun). Usage: tools/profile_bench.sh <tag> [bench args]
# Writes gpurun_out/prof_<tag>/{stats,pmc_*}/...; copy the summaries you want to keep into profiles/.
TAG=${1:-run}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-precision --no-end-to-end --no-sweep $@"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$C -o pmc -- $BENCH > $OUT/pmc_$C.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_lds -o pmc -- $BENCH > $OUT/pmc_lds.log 2>&1
rocprofv3 -L > $OUT/counters_list.txt 2>&1
find $OUT -name "*.csv" | head -50
