# Round-4 evidence session on one GPU box:   gpurun -- "bash tools/evidence_run_r04.sh"   -> gpurun_out/ev6, summarised into profiles/r04_*
# (python tools/evidence_r04_to_profiles.py).  Every profiled program stands directly after `--`; --pmc passes carry no other trace domain.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev6; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_f32.json 2> $O/bench_f32.err
echo bench done
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o b -- python3 $R/bench.py --no-cpu-baseline --no-multi-stream-region --steps 7 --warmup 3 > $O/trace_bench.json 2> $O/trace_bench.err
rocprofv3 --kernel-trace --output-format csv -d $O/trace_spade_bf16_act16 -o b -- python3 $R/bench.py --no-cpu-baseline --no-multi-stream --no-multi-stream-region --decoder spade --dtype bf16 --act16 --steps 4 --warmup 3 > $O/trace_spade_bf16_act16_bench.json 2> $O/trace_spade_bf16_act16_bench.err
echo traces done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --no-cpu-baseline --no-conv-timer --no-multi-stream-region --steps 2 --warmup 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --no-cpu-baseline --no-conv-timer --no-multi-stream-region --steps 2 --warmup 1 > /dev/null 2> $O/pmc_write.err
echo pmc traffic done
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --multi-stream > $O/bench_f32_multistream.json 2> $O/bench_f32_multistream.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 --act16 > $O/bench_bf16_act16.json 2> $O/bench_bf16_act16.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --decoder spade > $O/bench_spade_f32.json 2> $O/bench_spade_f32.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --decoder spade --dtype bf16 --act16 > $O/bench_spade_bf16_act16.json 2> $O/bench_spade_bf16_act16.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --decoder spade --dtype bf16 --act16 --no-multi-stream > $O/bench_spade_bf16_act16_onestream.json 2> $O/bench_spade_bf16_act16_onestream.err
python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --model mmsdnet --modalities 3 --size 320 --batch 16 --dtype f16 --act16 > $O/bench_mmsdnet3_320_f16_act16.json 2> $O/bench_mmsdnet3_320_f16_act16.err
python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --l_mix 0.1 > $O/bench_lmix01.json 2> $O/bench_lmix01.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --decoder spade --dtype bf16 --act16 --conv16 0 > $O/bench_spade_bf16_act16_conv16off.json 2> $O/bench_spade_bf16_act16_conv16off.err
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --conv16 0 > $O/bench_f32_conv16off.json 2> $O/bench_f32_conv16off.err
DTYPE=f32 timeout -k 10 400 python3 $R/tools/conv16_bench.py > $O/conv16_ab_f32.txt 2>&1
timeout -k 10 400 python3 $R/tools/conv16_bench.py > $O/conv16_ab.txt 2>&1
echo benches done
cd $R
python3 tools/kernel_stats.py $(find $O/trace -name "*kernel_trace.csv" | head -n 1) 10 "rocprofv3 --kernel-trace --output-format csv : python3 bench.py --no-cpu-baseline --no-multi-stream-region --steps 7 --warmup 3 (round 4 final, DAFNet-FiLM 256x256 bs8 fp32, 1 x MI355X)" > $O/final_kernel_stats.txt
python3 tools/kernel_stats.py $(find $O/trace_spade_bf16_act16 -name "*kernel_trace.csv" | head -n 1) 7 "rocprofv3 --kernel-trace : python3 bench.py --no-cpu-baseline --no-multi-stream --decoder spade --dtype bf16 --act16 --steps 4 --warmup 3 (round 4, BASELINE config #3 model)" > $O/kernel_stats_spade_bf16_act16.txt
python3 tools/gpu_busy.py $(find $O/trace -name "*kernel_trace.csv" | head -n 1) 10 3 > $O/gpu_busy.txt 2>&1 || true
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/conv_traffic.json dafnet-film-256-bs8-f32-lmix1 > $O/conv_traffic_families.txt

rm -rf $O/trace $O/trace_spade_bf16_act16 $O/pmc_fetch $O/pmc_write
echo summaries done
