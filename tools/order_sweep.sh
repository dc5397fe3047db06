#!/bin/bash
# How the band-ordered tile list is dealt to the XCDs (HGI_XCD_MODE: 0 contiguous eighths, 1 bands round-robin) x band
# height (HGI_ENC_BAND / HGI_DEC_BAND), on C4 (one 16384^2 frame, level 8) and on the C3 shard (64 x 4096^2, level 4):
# tools/c4_time.py once per combination (the library reads the switches once per process).  -> profiles/r03_order_sweep.txt
# The switches below exist in the KNOBS build of the library only (make -C rustyhgi_amd/csrc knobs; csrc/hgi_knobs.h):
# the release libhgi_hip.so reads nothing from the environment.
export HGI_LIB_PATH=${HGI_LIB_PATH:-$PWD/rustyhgi_amd/libhgi_hip_knobs.so}
run() { echo "HGI_XCD_MODE=$1 band=$2 $3"; env HGI_XCD_MODE=$1 HGI_ENC_BAND=$2 HGI_DEC_BAND=$2 $3 python tools/c4_time.py 2>/dev/null | grep " L[0-9]"; }
echo "C4: 16384^2 level 8 High, encode then decode (bench pattern), us per call"
for m in 0 1; do for b in 2 3 4 6 8 16; do run $m $b ""; done; done
echo "C3 shard: 64 x 4096^2 level 4"
for m in 0 1; do for b in 2 4 8 16; do run $m $b "C4_FRAMES=64 C4_SIZE=4096 C4_LEVELS=4"; done; done
