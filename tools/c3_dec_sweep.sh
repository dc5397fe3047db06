#!/bin/bash
# Decoder launch policies against the size of the batch: resident tiles per CU (HGI_DEC_WAVES: 0 = the 32 its LDS allows) x how the
# bands are dealt to the XCDs (HGI_XCD_MODE: 1 round-robin, 0 contiguous eighths), F x 4096^2 level 4, tools/c4_time.py once per
# setting on the KNOBS build.  -> profiles/r04_c3_dec_sweep.txt
# The switches below exist in the KNOBS build of the library only (make -C rustyhgi_amd/csrc knobs; csrc/hgi_knobs.h):
# the release libhgi_hip.so reads nothing from the environment.
export HGI_LIB_PATH=${HGI_LIB_PATH:-$PWD/rustyhgi_amd/libhgi_hip_knobs.so}
run() { echo "# $1"; env $1 C4_SIZE=4096 C4_LEVELS=4 python tools/c4_time.py 2>>${TRACE_FILE:-/dev/null} | grep " L[0-9]" | sed 's/ | grid.*//'; }
echo "tools/c3_dec_sweep.sh: F x 4096^2 level 4, encode then decode (bench pattern), us per call"
for f in ${FRAMES:-64 512}; do for m in 1 0; do for w in 0 24 20 16; do run "C4_FRAMES=$f HGI_XCD_MODE=$m HGI_DEC_WAVES=$w"; done; done; done
