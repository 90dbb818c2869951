#!/bin/bash
# LDS bank conflicts / instruction mix of the reduced-precision convolution kernels (rocprofv3 PMC; run from the repo root on a GPU box)
set -e
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/pmc_lp; mkdir -p $O; cd /tmp
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
export DTYPE=bf16
IO=0 rocprofv3 --pmc $C --output-format csv -d $O/wg_io0 -o p -- python3 $R/tools/wgrad_one.py 128 128 0 128 0 > /dev/null 2>&1
IO=5 rocprofv3 --pmc $C --output-format csv -d $O/wg_io5 -o p -- python3 $R/tools/wgrad_one.py 128 128 0 128 0 > /dev/null 2>&1
IO=0 rocprofv3 --pmc $C --output-format csv -d $O/fw_io0 -o p -- python3 $R/tools/conv_one.py 128 128 128 fwd > /dev/null 2>&1
IO=5 rocprofv3 --pmc $C --output-format csv -d $O/fw_io5 -o p -- python3 $R/tools/conv_one.py 128 128 128 fwd > /dev/null 2>&1
cd $R
for d in wg_io0 wg_io5 fw_io0 fw_io5; do echo "== $d"; python tools/pmc_summary.py $O/$d conv_; done
