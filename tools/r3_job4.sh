#!/bin/bash
# round 3, GPU session 4: three-stage seed prefetch, pinned requests, conflict-free split activation layout
TAG=${1:-r3d}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B=tools/ubench/bin
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q --timeout=240 -k "broyden3 or tracer_against or forward_against or sdf_eval or skin_lbs or joint_root or shade_composite" > $OUT/tests_canon.log 2>&1
echo "quick tests rc=$?"; tail -3 $OUT/tests_canon.log
timeout 1200 python tools/abn.py --rounds 2 base=- nopin=$B/libarah_nopin.so pin0=$B/libarah_pin0.so lo1hi1=$B/libarah_lo1hi1.so hi1=$B/libarah_hi1.so hi1nopin=$B/libarah_hi1nopin.so tile=-,ARAH_CANON_KERNEL=tile 2>&1 | tee $OUT/abn.txt
python tools/phase_clocks.py run 3 > $OUT/phase_clocks_wave.txt 2>&1; tail -10 $OUT/phase_clocks_wave.txt
timeout 900 python -m pytest tests -m gpu -x -q --timeout=240 > $OUT/tests.log 2>&1
echo "pytest rc=$?" >> $OUT/tests.log
tail -4 $OUT/tests.log
