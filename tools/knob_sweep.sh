#!/bin/bash
# One-box sweep of environment knobs on the C1 bench: default run between every variant; prints the step time and the
# kernel kind the knob targets.   tools/knob_sweep.sh <kernel-kind> VAR=v1 VAR=v2 ...
cd $GRAFT_REPO_ROOT
kind=$1; shift
run() { env "$@" python3 bench.py --steps 25 --warmup 6 --no-cpu-baseline --breakdown 2>&1 >/dev/null | grep -E "bench\] gpu|^$kind " | awk '{printf "%s ", ($1=="[bench]") ? $(NF-1) " ms/step" : $1 " " $3 " ms"} END {print ""}'; }
for v in "$@"; do echo "default      : $(run IMT_NOP=1)"; echo "$v : $(run $v)"; done
