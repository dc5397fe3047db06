#!/bin/bash
# The literal C3 batch on one GPU (512 x 4096^2, three planes of 8 GiB) against its shards: does a launch's time per frame
# depend on how many frames it holds, and does the decoder -- whose per-frame time grows with the batch, the encoder's does
# not -- respond to the launch policies at that size?  tools/c4_time.py once per setting on the KNOBS build.
#   -> profiles/r04_c3_big_sweep.txt
# The switches below exist in the KNOBS build of the library only (make -C rustyhgi_amd/csrc knobs; csrc/hgi_knobs.h):
# the release libhgi_hip.so reads nothing from the environment.
export HGI_LIB_PATH=${HGI_LIB_PATH:-$PWD/rustyhgi_amd/libhgi_hip_knobs.so}
run() { echo "# $1"; env $1 C4_SIZE=4096 C4_LEVELS=4 python tools/c4_time.py 2>>${TRACE_FILE:-/dev/null} | grep " L[0-9]" | sed 's/ | grid.*//'; }
echo "tools/c3_big_sweep.sh: F x 4096^2 level 4 High-table, encode then decode (bench pattern), us per call"
for f in 64 128 256 512; do run "C4_FRAMES=$f"; done
echo "512 frames, decoder policies"
for k in "HGI_DEC_WAVES=24" "HGI_DEC_WAVES=20" "HGI_DEC_WAVES=16" "HGI_DEC_BAND=2" "HGI_DEC_BAND=8" "HGI_XCD_MODE=0" "HGI_DEC_REVERSE=1"; do run "C4_FRAMES=512 $k"; done
