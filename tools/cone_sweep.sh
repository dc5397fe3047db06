#!/bin/bash
# The cone (seeds rebuilt in the tile kernel at four fused levels) against the older chains, same box, one process per row:
#   tools/cone_sweep.sh  ->  C4 (16384^2 L8), the 64 x 4096^2 batch and a lone 4096^2 frame at levels 5 ... 8
set -u
export C4_PLANE_BYTES=$((1<<30))
echo "## C4: 16384^2 level 8 High (tools/c4_time.py), placed planes"
for mode in "HGI_CONE=0" "HGI_CONE=1"; do
    echo "# $mode"; env $mode python3 tools/c4_time.py
done
for lv in 5 6 7 8; do
    echo "## 64 x 4096^2 level $lv"
    for mode in "HGI_CONE=0" "HGI_CONE_MIN_ENC=5 HGI_CONE_MIN_DEC=5"; do
        echo "# $mode"; env $mode C4_SIZE=4096 C4_FRAMES=64 C4_LEVELS=$lv C4_PLANE_BYTES=0 python3 tools/c4_time.py
    done
done
for lv in 5 6 7 8; do
    echo "## 1 x 4096^2 level $lv"
    for mode in "HGI_CONE=0" "HGI_CONE_MIN_ENC=5 HGI_CONE_MIN_DEC=5"; do
        echo "# $mode"; env $mode C4_SIZE=4096 C4_FRAMES=1 C4_LEVELS=$lv C4_PLANE_BYTES=0 python3 tools/c4_time.py
    done
done
