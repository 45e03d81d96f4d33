#!/bin/bash
# rocprofv3 PMC passes for the R1CS residual kernel (run on the GPU box through gpurun).
# usage: tools/prof_r1cs.sh <logn> <outdir>
set -u
LOGN=${1:-22}; OUT=${2:-gpurun_out/prof_r1cs}
mkdir -p $OUT; export TMPDIR=/tmp
CMD="python tools/kbench.py r1cs --logn $LOGN --reps 5 --copies 1"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/pmc_sq -o sq -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -o sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -30
