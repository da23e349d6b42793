#!/bin/bash
# Wave-state counters of the bench step per kernel (one PMC pass, counters only): where the waves of each kernel spend
# their cycles -- parked at s_waitcnt / barriers (SQ_WAIT_ANY), issue-stalled (SQ_WAIT_INST_ANY), or issuing
# (SQ_ACTIVE_INST_*).  Units are quad-cycles summed over waves (MI355X_MICROARCH.md, "rocprofv3 PMC slots").
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES \
  --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/sq.log 2>&1
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$OUT/sq/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:16]
print("%-70s %6s %8s | %6s %6s %6s | %6s %6s %6s" % ("kernel", "n", "wavecyc", "parked", "stall", "active", "valu", "lds", "vmem"))
for k, c in rows:
    w = c["SQ_WAVE_CYCLES"] or 1.0
    print("%-70s %6d %8.2e | %5.1f%% %5.1f%% %5.1f%% | %5.1f%% %5.1f%% %5.1f%%" % (k, n[k], w, 100*c["SQ_WAIT_ANY"]/w, 100*c["SQ_WAIT_INST_ANY"]/w,
          100*c["SQ_ACTIVE_INST_ANY"]/w, 100*c["SQ_ACTIVE_INST_VALU"]/w, 100*c["SQ_ACTIVE_INST_LDS"]/w, 100*c["SQ_ACTIVE_INST_VMEM"]/w))
PY
