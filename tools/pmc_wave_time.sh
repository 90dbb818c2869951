#!/bin/bash
# Where the waves of one convolution kernel spend their cycles (rocprofv3 PMC, two passes; run from the repo root on a GPU box):
#   tools/pmc_wave_time.sh <out tag> <conv_one.py | wgrad_one.py> <args...>      (DTYPE / IO from the environment)
set -e
export TMPDIR=/tmp
R=$(pwd); tag=$1; shift; prog=$1; shift
O=$R/gpurun_out/pmc_wave_$tag; mkdir -p $O; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/a -o p -- python3 $R/tools/$prog "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/b -o p -- python3 $R/tools/$prog "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAVES SQ_LEVEL_WAVES --output-format csv -d $O/c -o p -- python3 $R/tools/$prog "$@" > /dev/null 2>&1 || true
cd $R
for d in a b c; do python tools/pmc_summary.py $O/$d conv_; done
