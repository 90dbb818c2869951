#!/bin/bash
# ablation timings of conv16_kernel (measurement build libmmseg_hip_ab.so): bash tools/abl_conv16.sh B H C1 C2 Cout k ups
export MMSEG_HIP_LIB=$GRAFT_REPO_ROOT/multimodal_segmentation_amd/csrc/libmmseg_hip_ab.so
cd $GRAFT_REPO_ROOT
for abl in 0 1 2 3; do MMSEG_CONV16_ABL=$abl python3 tools/conv16_one.py 2 "$@" 30 2>&1 | grep mode; done
python3 tools/conv16_one.py 0 "$@" 30 2>&1 | grep mode
