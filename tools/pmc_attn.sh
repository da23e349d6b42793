#!/bin/bash
# SQ counters of the attention / LayerNorm-backward kernels (counters only, three passes): where the waves' cycles go, LDS bank
# conflicts, instruction counts.  Units: quad-cycles summed over waves (MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_attn
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES \
  --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/tools/attn_pmc_driver.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES \
  --output-format csv -d $OUT/p2 -- python3 $GRAFT_REPO_ROOT/tools/attn_pmc_driver.py > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p3 -- python3 $GRAFT_REPO_ROOT/tools/attn_pmc_driver.py > $OUT/p3.log 2>&1
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$OUT/p[12]/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES",): n[k] += 1
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:8]:
    w = c["SQ_WAVE_CYCLES"] or 1.0; m = max(1, n[k])
    print("%s  (n=%d)" % (k, m))
    print("   wave quad-cycles/launch %.3e  waves %d | parked %.1f%% stall %.1f%% active %.1f%% | valu %.1f%% lds %.1f%% vmem %.1f%%" % (
        w / m, c["SQ_WAVES"] / m, 100*c["SQ_WAIT_ANY"]/w, 100*c["SQ_WAIT_INST_ANY"]/w, 100*c["SQ_ACTIVE_INST_ANY"]/w,
        100*c["SQ_ACTIVE_INST_VALU"]/w, 100*c["SQ_ACTIVE_INST_LDS"]/w, 100*c["SQ_ACTIVE_INST_VMEM"]/w))
    print("   per launch: insts valu %.3e lds %.3e mfma %.3e | lds idx active %.3e bank conflict %.3e (%.1f%%) | mfma busy cyc %.3e | wait_inst_lds %.3e | busy cycles %.3e" % (
        c["SQ_INSTS_VALU"]/m, c["SQ_INSTS_LDS"]/m, c["SQ_INSTS_MFMA"]/m, c["SQ_LDS_IDX_ACTIVE"]/m, c["SQ_LDS_BANK_CONFLICT"]/m,
        100*c["SQ_LDS_BANK_CONFLICT"]/max(1.0,c["SQ_LDS_IDX_ACTIVE"]), c["SQ_VALU_MFMA_BUSY_CYCLES"]/m, c["SQ_WAIT_INST_LDS"]/m, c["SQ_BUSY_CYCLES"]/m))
PY
cat $(ls $OUT/p3/*/*kernel_stats.csv | head -1) | cut -c1-160 | head -8
