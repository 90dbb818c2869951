# One GPU-box session that regenerates the evidence kept under profiles/ (bench lines, kernel trace, PMC passes, Dice seeds):
#   gpurun -- "bash tools/evidence_run.sh"; summaries are then made from gpurun_out/ev2 with the tools/*.py scripts.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev2; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 3 > $O/bench_f32.json 2> $O/bench_f32.err
echo bench done
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o b -- python3 $R/bench.py --no-cpu-baseline --steps 7 --warmup 3 > $O/trace_bench.json 2> $O/trace_bench.err
echo trace done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --no-cpu-baseline --no-conv-timer --steps 2 --warmup 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --no-cpu-baseline --no-conv-timer --steps 2 --warmup 1 > /dev/null 2> $O/pmc_write.err
echo pmc traffic done
for shape in "fwd 128 128 128" "fwd 256 64 64"; do set -- $shape; n=$1_$2_$3_$4
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_$n -o p -- python3 $R/tools/conv_one.py $2 $3 $4 fwd > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_lds_$n -o p -- python3 $R/tools/conv_one.py $2 $3 $4 fwd > /dev/null 2>&1
done
for shape in "128 128 0 128 0" "256 64 0 64 0"; do set -- $shape; n=wgrad_$1_$2_$4
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_$n -o p -- python3 $R/tools/wgrad_one.py $1 $2 $3 $4 $5 > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_lds_$n -o p -- python3 $R/tools/wgrad_one.py $1 $2 $3 $4 $5 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$n -o p -- python3 $R/tools/wgrad_one.py $1 $2 $3 $4 $5 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$n -o p -- python3 $R/tools/wgrad_one.py $1 $2 $3 $4 $5 > /dev/null 2>&1
done
echo pmc kernels done
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --decoder spade > $O/bench_spade_f32.json 2> $O/bench_spade_f32.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --decoder spade --dtype bf16 > $O/bench_spade_bf16.json 2> $O/bench_spade_bf16.err
python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --model mmsdnet --modalities 3 --size 320 --batch 16 --dtype f16 > $O/bench_mmsdnet3_320_f16.json 2> $O/bench_mmsdnet3_320_f16.err
python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --l_mix 0.1 > $O/bench_lmix01.json 2> $O/bench_lmix01.err
echo benches done
for s in 0 1 2 3 4; do python3 $R/tools/dice_seeds.py $s product 500 64 4 1e-3 350 10 > $O/dice_seed${s}_product.log 2>&1; tail -1 $O/dice_seed${s}_product.log; done
