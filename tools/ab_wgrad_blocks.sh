export MMSEG_HIP_LIB=$GRAFT_REPO_ROOT/multimodal_segmentation_amd/csrc/libmmseg_hip_ab.so
cd $GRAFT_REPO_ROOT
for t in 256 512 768 1024; do echo "== target $t"; MMSEG_WGRAD32H_BLOCKS=$t python3 tools/wgrad32h_bench.py 2>&1 | grep "^(" | awk '{print $1,$2,$3,$4,$5,$6, $8, $10}'; done
