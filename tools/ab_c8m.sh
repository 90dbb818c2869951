# A/B of the 8 -> 8 weight gradient (VALU kernel vs 4x4x1 MFMA kernel) inside the steps that use it; measurement build (-DMMSEG_AB)
export MMSEG_HIP_LIB=$GRAFT_REPO_ROOT/multimodal_segmentation_amd/csrc/libmmseg_hip_ab.so
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for m in 0 1; do
echo "== MMSEG_WGRAD_C8_MFMA=$m (rep $rep)"
MMSEG_WGRAD_C8_MFMA=$m python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-multi-stream-region --model mmsdnet --modalities 3 --size 320 --batch 16 --dtype f16 --act16 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('mmsdnet3 f16', d['value'], d['ms_per_step'])"
MMSEG_WGRAD_C8_MFMA=$m python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-multi-stream-region --dtype bf16 --act16 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('film bf16', d['value'], d['ms_per_step'])"
done
done
