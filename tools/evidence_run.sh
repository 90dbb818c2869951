# One GPU-box session that regenerates the evidence kept under profiles/ (bench lines, kernel traces, GPU-busy with / without hipGraphs,
# PMC passes):   gpurun -- "bash tools/evidence_run.sh"; summaries are then made from gpurun_out/ev4 with the tools/*.py scripts
# (DESIGN.md section 6).  Every profiled program stands directly after `--`; --pmc passes carry no other trace domain.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev4; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 3 > $O/bench_f32.json 2> $O/bench_f32.err
echo bench done
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o b -- python3 $R/bench.py --no-cpu-baseline --steps 7 --warmup 3 > $O/trace_bench.json 2> $O/trace_bench.err
rocprofv3 --kernel-trace --output-format csv -d $O/trace_graphs -o b -- python3 $R/bench.py --no-cpu-baseline --graphs --steps 7 --warmup 3 > $O/trace_graphs_bench.json 2> $O/trace_graphs_bench.err
rocprofv3 --kernel-trace --output-format csv -d $O/trace_bf16_act16 -o b -- python3 $R/bench.py --no-cpu-baseline --dtype bf16 --act16 --steps 7 --warmup 3 > $O/trace_bf16_act16_bench.json 2> $O/trace_bf16_act16_bench.err
rocprofv3 --kernel-trace --output-format csv -d $O/trace_spade_bf16_act16 -o b -- python3 $R/bench.py --no-cpu-baseline --decoder spade --dtype bf16 --act16 --steps 4 --warmup 3 > $O/trace_spade_bf16_act16_bench.json 2> $O/trace_spade_bf16_act16_bench.err
echo traces done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --no-cpu-baseline --no-conv-timer --steps 2 --warmup 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --no-cpu-baseline --no-conv-timer --steps 2 --warmup 1 > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_bf16a -o p -- python3 $R/bench.py --no-cpu-baseline --no-conv-timer --dtype bf16 --act16 --steps 2 --warmup 1 > /dev/null 2> $O/pmc_fetch_bf16a.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_bf16a -o p -- python3 $R/bench.py --no-cpu-baseline --no-conv-timer --dtype bf16 --act16 --steps 2 --warmup 1 > /dev/null 2> $O/pmc_write_bf16a.err
echo pmc traffic done
for shape in "fwd 128 128 128" "fwd 256 64 64"; do set -- $shape; n=$1_$2_$3_$4
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_$n -o p -- python3 $R/tools/conv_one.py $2 $3 $4 fwd > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_lds_$n -o p -- python3 $R/tools/conv_one.py $2 $3 $4 fwd > /dev/null 2>&1
done
for shape in "128 128 0 128 0" "256 64 0 64 0"; do set -- $shape; n=wgrad_$1_$2_$4
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_$n -o p -- python3 $R/tools/wgrad_one.py $1 $2 $3 $4 $5 > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_lds_$n -o p -- python3 $R/tools/wgrad_one.py $1 $2 $3 $4 $5 > /dev/null 2>&1
done
echo pmc fp32 kernels done
export DTYPE=bf16
for io in 0 5; do export IO=$io
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc16_mfma_fwd_io$io -o p -- python3 $R/tools/conv_one.py 128 128 128 fwd > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $O/pmc16_lds_fwd_io$io -o p -- python3 $R/tools/conv_one.py 128 128 128 fwd > /dev/null 2>&1
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc16_mfma_wgrad_io$io -o p -- python3 $R/tools/wgrad_one.py 128 128 0 128 0 > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $O/pmc16_lds_wgrad_io$io -o p -- python3 $R/tools/wgrad_one.py 128 128 0 128 0 > /dev/null 2>&1
done
unset DTYPE IO
echo pmc 16-bit kernels done
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --graphs > $O/bench_f32_graphs.json 2> $O/bench_f32_graphs.err
python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --multi-stream > $O/bench_f32_multistream.json 2> $O/bench_f32_multistream.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --multi-stream --dtype bf16 --act16 > $O/bench_bf16_act16_multistream.json 2> $O/bench_bf16_act16_multistream.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --multi-stream --decoder spade --dtype bf16 --act16 > $O/bench_spade_bf16_act16_multistream.json 2> $O/bench_spade_bf16_act16_multistream.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 --act16 > $O/bench_bf16_act16.json 2> $O/bench_bf16_act16.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --decoder spade > $O/bench_spade_f32.json 2> $O/bench_spade_f32.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --decoder spade --dtype bf16 > $O/bench_spade_bf16.json 2> $O/bench_spade_bf16.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --decoder spade --dtype bf16 --act16 > $O/bench_spade_bf16_act16.json 2> $O/bench_spade_bf16_act16.err
python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --model mmsdnet --modalities 3 --size 320 --batch 16 --dtype f16 > $O/bench_mmsdnet3_320_f16.json 2> $O/bench_mmsdnet3_320_f16.err
python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --model mmsdnet --modalities 3 --size 320 --batch 16 --dtype f16 --act16 > $O/bench_mmsdnet3_320_f16_act16.json 2> $O/bench_mmsdnet3_320_f16_act16.err
python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --l_mix 0.1 > $O/bench_lmix01.json 2> $O/bench_lmix01.err
echo benches done
