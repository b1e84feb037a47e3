#!/bin/bash
TAG=${1:-r3m}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B=tools/ubench/bin
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
{
for T in 128 64; do
  ARAH_DENSITY_TILE=$T python tools/ablate_trunk.py --density
  for v in ahead2 pp pp_ahead2; do
    ARAH_DENSITY_TILE=$T ARAH_LIB_PATH=$B/libarah_$v.so python tools/ablate_trunk.py --density
  done
done
} 2>&1 | grep -v "amdgpu.ids\|Warning" | tee $OUT/trunk_variants.txt
