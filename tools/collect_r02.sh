#!/bin/bash
# Round-2 evidence set (run on the GPU box through gpurun; the summaries are copied into profiles/ by hand afterwards).
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
rm -rf $O; mkdir -p $O
bash tools/collect_round_profile.sh r02 > $O/collect.log 2>&1
cp gpurun_out/prof_r02/* $O/ 2>/dev/null || true
bash tools/collect_traffic.sh > $O/traffic.log 2>&1
python3 tools/make_traffic_json.py gpurun_out/pmc_traffic.json 2 > $O/traffic_table.txt 2>&1
cp profiles/r02_pmc_traffic.json $O/
for c in c1v60k c1ragged ref768; do python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null; done > $O/side_configs.json
python3 tools/bench_aux.py --breakdown > $O/aux_workloads.json 2> $O/aux_breakdown.txt
python3 tools/beam_bench.py > $O/beam_search.txt 2>&1
python3 tools/gemm_ln_bench.py 2>&1 | grep -v amdgpu > $O/gemm_ln_study.txt
IMT_TRACE=gemm_ln python3 tools/gemm_ln_trace.py 2>&1 | grep -v amdgpu >> $O/gemm_ln_study.txt
python3 tools/launch_cost.py 2>&1 | grep -v amdgpu > $O/launch_cost.txt
bash tools/collect_gaps.sh > $O/gaps.txt 2>&1
hipcc --offload-arch=gfx950 -O3 tools/probe_soffset.hip -o /tmp/ps && /tmp/ps > $O/probe_soffset.txt 2>&1
hipcc --offload-arch=gfx950 -O3 tools/probe_fill.hip -o /tmp/pf && timeout -k 10 200 /tmp/pf 2 > $O/probe_fill_2mib.txt 2>&1
bash tools/ab_adam.sh > $O/adam_overlap_ab.txt 2>&1
tail -1 $O/bench.json | cut -c1-300
