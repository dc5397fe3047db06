#!/bin/bash
# How the band-ordered tile list is dealt to the XCDs (0: contiguous eighths, 1: whole bands round-robin) against the size of the
# batch: F x 4096^2, level 4, tools/c4_time.py once per setting on the KNOBS build (HGI_XCD_MODE).  -> profiles/r04_c3_xcd_sweep.txt
# The switches below exist in the KNOBS build of the library only (make -C rustyhgi_amd/csrc knobs; csrc/hgi_knobs.h):
# the release libhgi_hip.so reads nothing from the environment.
export HGI_LIB_PATH=${HGI_LIB_PATH:-$PWD/rustyhgi_amd/libhgi_hip_knobs.so}
run() { echo "# $1"; env $1 C4_SIZE=4096 C4_LEVELS=4 python tools/c4_time.py 2>>${TRACE_FILE:-/dev/null} | grep " L[0-9]" | sed 's/ | grid.*//'; }
echo "tools/c3_xcd_sweep.sh: F x 4096^2 level 4, encode then decode (bench pattern), us per call"
for f in ${FRAMES:-32 64 96 128 192 256 384 512}; do for m in 1 0; do run "C4_FRAMES=$f HGI_XCD_MODE=$m"; done; done
