#!/bin/bash
# A/B of one environment switch on the C1 bench inside one box: tools/ab_env.sh VAR=value   (interleaved: default, switch, default, switch)
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_env; mkdir -p $O
for i in 1 2; do
  python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --breakdown > $O/base_$i.json 2> $O/base_$i.txt || exit 1
  echo "default $i $(grep 'bench\] gpu' $O/base_$i.txt)"
  env "$@" python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --breakdown > $O/var_$i.json 2> $O/var_$i.txt || exit 1
  echo "$* $i $(grep 'bench\] gpu' $O/var_$i.txt)"
done
