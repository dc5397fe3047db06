#!/bin/bash
export HGI_LIB_PATH=$PWD/rustyhgi_amd/libhgi_hip_knobs.so
run() { echo "# $1"; env $1 C4_SIZE=4096 C4_LEVELS=4 python tools/c4_time.py 2>/dev/null | grep " L[0-9]" | sed 's/ | grid.*//' | cut -c1-125; }
for f in 512 64; do for w in 10 12 14 16 18 20; do run "C4_FRAMES=$f HGI_DEC_WAVES=$w"; done; done
for w in 0 18 16 14; do run "C4_FRAMES=512 HGI_ENC_WAVES=$w"; done
