export MMSEG_HIP_LIB=$GRAFT_REPO_ROOT/multimodal_segmentation_amd/csrc/libmmseg_hip_ab.so
cd $GRAFT_REPO_ROOT
for shape in "8 128 128 0 128 3 0" "8 128 128 128 128 3 0" "48 64 128 0 128 3 0" "8 256 64 0 64 3 0" "8 256 64 64 64 3 0"; do
for sup in 1 2; do MMSEG_CONV16H_SUP=$sup python3 tools/conv16_one.py 2 $shape 30 2>&1 | grep mode | sed "s/^/sup $sup /"; done
python3 tools/conv16_one.py 0 $shape 30 2>&1 | grep mode
done
