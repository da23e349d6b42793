#!/bin/bash
# A/B of builds of the library on the C1 bench inside one box: tools/ab_lib.sh <variant> [<variant> ...]
# (variants built by tools/build_variant.sh).  Two interleaved rounds: base, v1, v2, ..., base, v1, v2, ...
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O
for i in 1 2; do
  for v in base "$@"; do
    lib=$PWD/build/libimt_hip_$v.so; [ $v = base ] && lib=$PWD/imagetranslate_amd/libimt_hip.so
    IMT_PROF_SHAPES=1 IMT_LIB=$lib python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --breakdown > $O/${v}_$i.json 2> $O/${v}_$i.txt || { tail -5 $O/${v}_$i.txt; exit 1; }
    echo "$v $i $(python3 -c "import json; print(json.loads(open('$O/${v}_$i.json').read().strip().splitlines()[-1])['ms_per_step'])")"
  done
done
