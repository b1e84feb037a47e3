#!/bin/bash
# round 3, GPU session 3: new tail (T on the matrix pipe, joint Broyden), one-N-tile variants with 3 waves per SIMD
TAG=${1:-r3c}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B=tools/ubench/bin
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
for L in "" $B/libarah_nt1.so; do
  ARAH_LIB_PATH=$L timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q --timeout=240 -k "broyden3 or tracer_against or forward_against" > $OUT/tests_canon.log 2>&1
  echo "canon tests [$L] rc=$?"; tail -3 $OUT/tests_canon.log
done
timeout 1200 python tools/abn.py --rounds 2 base=- hi0=$B/libarah_hi0.so nt1=$B/libarah_nt1.so nt1hi0=$B/libarah_nt1hi0.so tile=-,ARAH_CANON_KERNEL=tile 2>&1 | tee $OUT/abn.txt
python tools/phase_clocks.py run 3 > $OUT/phase_clocks_wave.txt 2>&1; tail -10 $OUT/phase_clocks_wave.txt
