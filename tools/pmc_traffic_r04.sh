# HBM traffic per convolution launch of the headline workload (round 4): two separate --pmc passes, no other trace domain
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev6; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --no-cpu-baseline --no-conv-timer --no-multi-stream-region --steps 2 --warmup 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --no-cpu-baseline --no-conv-timer --no-multi-stream-region --steps 2 --warmup 1 > /dev/null 2> $O/pmc_write.err
cd $R
rm -f $O/conv_traffic.json
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/conv_traffic.json dafnet-film-256-bs8-f32-lmix1 > $O/conv_traffic_families.txt
rm -rf $O/pmc_fetch $O/pmc_write
grep -c conv16h $O/conv_traffic.json
