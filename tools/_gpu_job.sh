cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "nearest or tracer or forward_against or reproducible" 2>&1 | tail -3
python tools/ab.py tools/ubench/bin/libarah_prev.so arah_release_amd/libarah_hip.so 2 2>&1 | grep -v amdgpu.ids
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2w; mkdir -p $OUT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --passes default > $OUT/bench_prof.json 2> $OUT/prof.err
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/prof -name "*.db" | head -1); python tools/rocpd_stats.py $DB | grep "sort_verts\|cell_clusters\|total"; rm -rf $OUT/prof
