#!/bin/bash
# L2 hit rate per kernel of the bench step (one PMC pass, counters only): TCC_HIT / (TCC_HIT + TCC_MISS), MI355X_MICROARCH.md "L2".
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_l2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p.log 2>&1
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$OUT/p/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:64]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "TCC_HIT_sum": n[k] += 1
for k, c in sorted(agg.items(), key=lambda kv: -(kv[1]["TCC_HIT_sum"] + kv[1]["TCC_MISS_sum"]))[:18]:
    h, m = c["TCC_HIT_sum"], c["TCC_MISS_sum"]
    print("%-64s n=%4d  hit %5.1f%%  requests/launch %.3e (x128 B = %.1f MB)" % (k, n[k], 100 * h / max(1.0, h + m), (h + m) / max(1, n[k]), (h + m) / max(1, n[k]) * 128 / 1e6))
PY
