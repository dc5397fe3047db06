#!/bin/bash
# Band height of the interior tile walk (fast_tile: bands of R tile rows, column-major inside a band) on BASELINE
# config C4 and on an 8192-wide frame: tools/c4_time.py once per value of HGI_DEC_BAND / HGI_ENC_BAND (the library reads
# them once per process).  Output -> profiles/r03_band_sweep.txt
# The switches below exist in the KNOBS build of the library only (make -C rustyhgi_amd/csrc knobs; csrc/hgi_knobs.h):
# the release libhgi_hip.so reads nothing from the environment.
export HGI_LIB_PATH=${HGI_LIB_PATH:-$PWD/rustyhgi_amd/libhgi_hip_knobs.so}
echo "tools/band_sweep.sh: 16384^2 level 8 High, encode then decode (bench pattern), us per call; default policy first"
python tools/c4_time.py 2>/dev/null | grep L8
for b in 1 2 3 4 6 8 16; do echo "HGI_DEC_BAND=$b HGI_ENC_BAND=$b"; HGI_DEC_BAND=$b HGI_ENC_BAND=$b python tools/c4_time.py 2>/dev/null | grep L8; done
echo "8192^2 level 7 High"
C4_SIZE=8192 C4_LEVELS=7 python tools/c4_time.py 2>/dev/null | grep L7
for b in 2 4 8 16; do echo "HGI_DEC_BAND=$b HGI_ENC_BAND=$b"; C4_SIZE=8192 C4_LEVELS=7 HGI_DEC_BAND=$b HGI_ENC_BAND=$b python tools/c4_time.py 2>/dev/null | grep L7; done
