#!/bin/bash
# One round's profile set (run on the GPU box through gpurun; copy what is to be judged into profiles/):
#   bench.json                 default bench.py run (with the CPU baseline leg)
#   kernel_breakdown.txt       per-kernel-kind table of the in-process launch timer (HIP events per launch)
#   per_shape_breakdown.txt    the same with GEMM kinds split by shape
#   kernel_stats.csv           rocprofv3 --kernel-trace --stats of the same command
#   bench_under_rocprof.json   the bench line printed by that profiled run
set -e
TAG=${1:-cur}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --steps 20 --warmup 5 --breakdown --no-cpu-baseline > /dev/null 2> $OUT/kernel_breakdown.raw
IMT_PROF_SHAPES=1 python3 bench.py --steps 20 --warmup 5 --breakdown --no-cpu-baseline > /dev/null 2> $OUT/per_shape_breakdown.raw
grep -v amdgpu $OUT/kernel_breakdown.raw > $OUT/kernel_breakdown.txt
grep -v amdgpu $OUT/per_shape_breakdown.raw > $OUT/per_shape_breakdown.txt
rm -f $OUT/*.raw
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rp.err
cp $(ls $OUT/rp/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/rp
tail -1 $OUT/bench.json | cut -c1-330
