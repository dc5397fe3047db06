#!/bin/bash
# Resident tiles per CU (dynamic LDS padded: HGI_DEC_WAVES / HGI_ENC_WAVES) on C4 and on the C3 shard: does a launch that is
# only a few rounds deep gain from a shorter tile lifetime (fewer resident tiles at the same rate)?  -> profiles/r03_waves_sweep.txt
# The switches below exist in the KNOBS build of the library only (make -C rustyhgi_amd/csrc knobs; csrc/hgi_knobs.h):
# the release libhgi_hip.so reads nothing from the environment.
export HGI_LIB_PATH=${HGI_LIB_PATH:-$PWD/rustyhgi_amd/libhgi_hip_knobs.so}
run() { echo "waves: decode $1 encode $2 $3"; env HGI_DEC_WAVES=$1 HGI_ENC_WAVES=$2 $3 python tools/c4_time.py 2>/dev/null | grep " L[0-9]" | sed 's/ | grid.*//'; }
echo "C4: 16384^2 level 8 High, encode then decode (bench pattern), us per call (0 = what LDS and registers allow: 32 / 20)"
for w in "0 0" "24 16" "20 14" "16 12" "12 10" "10 8"; do run $w ""; done
echo "C3 shard: 64 x 4096^2 level 4"
for w in "0 0" "24 16" "16 12"; do run $w "C4_FRAMES=64 C4_SIZE=4096 C4_LEVELS=4"; done
