export MMSEG_HIP_LIB=$GRAFT_REPO_ROOT/multimodal_segmentation_amd/csrc/libmmseg_hip_ab.so
cd $GRAFT_REPO_ROOT
for dt in f32 bf16; do
for shape in "8 256 64 0 64 3 0" "8 256 64 64 64 3 0" "8 256 128 0 64 3 1"; do
for th in 0 1; do DTYPE=$dt MMSEG_CONV16H_N64_TH8=$th python3 tools/conv16_one.py 2 $shape 30 2>&1 | grep mode | sed "s/^/$dt n64_th8 $th /"; done
done; done
