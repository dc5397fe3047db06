#!/bin/bash
for lv in 5 6; do
echo "## 64 x 4096^2 L$lv: _nc = no cone, _c5 = cone"; AB_LEVELS=$lv python3 tools/ab.py -r 5 -s 12 _nc _c5 2>&1 | grep "variant\[" | grep -v fingerprint
done
echo "## 1 x 4096^2 L6 (C2 shape)"; AB_F=1 AB_LEVELS=6 AB_QUANT=0 python3 tools/ab.py -r 9 -s 40 _nc _c5 2>&1 | grep "variant\[" | grep -v fingerprint
echo "## 16 x 1920x1080 L6"; AB_F=16 AB_W=1920 AB_H=1080 AB_LEVELS=6 python3 tools/ab.py -r 9 -s 40 _nc _c5 2>&1 | grep "variant\[" | grep -v fingerprint
for lv in 5 6 7 8; do
echo "## size sweep L$lv default"; python3 tools/size_sweep.py $lv 2>&1 | grep " x "
echo "## size sweep L$lv HGI_CONE_MIN=5"; HGI_CONE_MIN_ENC=5 HGI_CONE_MIN_DEC=5 python3 tools/size_sweep.py $lv 2>&1 | grep " x "
done
