#!/bin/bash
export HGI_LIB_PATH=$PWD/rustyhgi_amd/libhgi_hip_tl.so
echo "### cone"; python3 tools/timeline.py c4 2>&1 | grep -v amdgpu.ids
echo "### HGI_CONE=0"; HGI_CONE=0 python3 tools/timeline.py c4 2>&1 | grep -v amdgpu.ids
