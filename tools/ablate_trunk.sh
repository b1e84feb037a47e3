#!/bin/bash
# the ablation ladder of the forward SDF trunk on one box -> stdout (profiles/r03_trunk_ablation.txt)
B=tools/ubench/bin
python tools/ablate_trunk.py
for v in NO_EPI NO_BARRIER A_FIXED B_FIXED AB_FIXED ONE_MFMA ALL; do
  ARAH_LIB_PATH=$B/libarah_abl_$v.so python tools/ablate_trunk.py
done
for T in 128 64; do
  ARAH_DENSITY_TILE=$T python tools/ablate_trunk.py --density
  for v in NO_EPI NO_BARRIER A_FIXED B_FIXED AB_FIXED ONE_MFMA ALL; do
    ARAH_DENSITY_TILE=$T ARAH_LIB_PATH=$B/libarah_abl_$v.so python tools/ablate_trunk.py --density
  done
done
