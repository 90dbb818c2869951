export MMSEG_HIP_LIB=$GRAFT_REPO_ROOT/multimodal_segmentation_amd/csrc/libmmseg_hip_ab.so
cd $GRAFT_REPO_ROOT
for shape in "8 128 64 0 128 3 0" "8 128 128 0 128 3 0" "8 128 128 128 128 3 0" "8 128 256 0 128 3 1"; do
for th in 0 1; do DTYPE=f32 MMSEG_CONV16H_TH16=$th python3 tools/conv16_one.py 2 $shape 20 2>&1 | grep mode | sed "s/^/f32 th16 $th /"; done
DTYPE=f32 python3 tools/conv16_one.py 0 $shape 20 2>&1 | grep mode | sed "s/^/f32 old /"
done
