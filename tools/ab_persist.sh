export MMSEG_HIP_LIB=$GRAFT_REPO_ROOT/multimodal_segmentation_amd/csrc/libmmseg_hip_ab.so
cd $GRAFT_REPO_ROOT
for dt in bf16 f32; do
for shape in "8 256 64 0 64 3 0" "8 256 64 64 64 3 0" "8 128 128 0 128 3 0" "8 128 128 128 128 3 0" "48 128 128 0 64 3 0" "48 32 128 0 256 3 0"; do
for pe in 0 1; do DTYPE=$dt MMSEG_CONV16H_PERSIST=$pe python3 tools/conv16_one.py 1 $shape 30 2>&1 | grep mode | sed "s/^/$dt persist $pe /"; done
done; done
