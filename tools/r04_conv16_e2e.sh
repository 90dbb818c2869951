# end-to-end effect of the large-tile 16-bit kernels (round 4): config #3 / #5 models with mmseg_conv16_mode 0 vs 1
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c16e2e; mkdir -p $O; cd $R
for m in 0 1; do
python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-multi-stream-region --decoder spade --dtype bf16 --act16 --conv16 $m > $O/spade_bf16_c$m.json 2> $O/spade_bf16_c$m.err
python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-multi-stream-region --dtype bf16 --act16 --conv16 $m > $O/film_bf16_c$m.json 2> $O/film_bf16_c$m.err
python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-multi-stream-region --model mmsdnet --modalities 3 --size 320 --batch 16 --dtype f16 --act16 --conv16 $m > $O/mmsdnet3_f16_c$m.json 2> $O/mmsdnet3_f16_c$m.err
echo mode $m done
done
python3 - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/c16e2e/*.json')):
    l = [x for x in open(f).read().splitlines() if x.startswith('{')]
    if l:
        d = json.loads(l[-1]); print(os.path.basename(f), d['value'], d['ms_per_step'], d.get('roofline', {}).get('kernel'), d.get('roofline', {}).get('frac'))
PY
