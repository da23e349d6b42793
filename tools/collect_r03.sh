#!/bin/bash
# Round-3 evidence set (run on the GPU box through gpurun; the summaries are copied into profiles/ by hand afterwards).
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
rm -rf $O; mkdir -p $O
bash tools/collect_round_profile.sh r03 > $O/collect.log 2>&1
cp gpurun_out/prof_r03/* $O/ 2>/dev/null || true
bash tools/collect_traffic.sh > $O/traffic.log 2>&1
python3 tools/make_traffic_json.py gpurun_out/pmc_traffic.json 3 > $O/traffic_table.txt 2>&1
cp profiles/r03_pmc_traffic.json $O/
for c in c1v60k c1ragged ref768; do python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null; done > $O/side_configs.json
python3 tools/bench_aux.py --breakdown > $O/aux_workloads.json 2> $O/aux_breakdown.raw
grep -v amdgpu $O/aux_breakdown.raw > $O/aux_breakdown.txt; rm -f $O/aux_breakdown.raw
python3 tools/beam_bench.py 2>&1 | grep -v amdgpu > $O/beam_search.txt
bash tools/pmc_attn.sh 2>&1 | grep -v amdgpu > $O/attn_ln_pmc.txt
python3 tools/attn_variants.py 2>&1 | grep -v amdgpu > $O/attn_variants.txt
bash tools/pmc_l2.sh 2>&1 | grep -v amdgpu > $O/l2_hit_rates.txt
python3 tools/overlap_probe6.py 2>&1 | grep -v amdgpu > $O/overlap_probe6.txt
python3 tools/tn_small_k.py 2>&1 | grep -v amdgpu > $O/tn_small_k.txt
(for i in 1 2; do for v in 0 1; do echo "IMT_GEMM_SHARE_CUS=$v  $(IMT_GEMM_SHARE_CUS=$v python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step")')"; done; done) > $O/share_cus_ab.txt
tail -1 $O/bench.json | cut -c1-300
