#!/bin/bash
# PMC passes over one 16-bit convolution shape with both kernels:  gpurun -- "bash tools/pmc_conv16.sh <tag> B H C1 C2 Cout k ups"
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
O=$R/gpurun_out/pmc16_$tag; mkdir -p $O; cd /tmp
for mode in 0 2; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/a$mode -o p -- python3 $R/tools/conv16_one.py $mode "$@" > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/b$mode -o p -- python3 $R/tools/conv16_one.py $mode "$@" > /dev/null 2>&1
  rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAVES SQ_LEVEL_WAVES --output-format csv -d $O/c$mode -o p -- python3 $R/tools/conv16_one.py $mode "$@" > /dev/null 2>&1 || true
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/d$mode -o p -- python3 $R/tools/conv16_one.py $mode "$@" > /dev/null 2>&1 || true
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$mode -o p -- python3 $R/tools/conv16_one.py $mode "$@" > /dev/null 2>&1 || true
done
cd $R
for mode in 0 2; do for d in a b c d; do echo "== mode $mode pass $d"; python3 tools/pmc_summary.py $O/$d$mode conv; done; grep -h "conv" $(find $O/t$mode -name "*kernel_stats.csv") | head -3; done > $O/summary.txt 2>&1
rm -rf $O/a? $O/b? $O/c? $O/d? $O/t?
cat $O/summary.txt
