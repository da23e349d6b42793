#!/bin/bash
# PMC passes (counters only: no kernel-trace/stats mixing) for one GEMM shape.  usage: pmc_gemm.sh TAG LAYOUT M N K [split]
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py "$@" > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $OUT/p3 -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for p in ("p1","p2","p3"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % p):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print("$TAG", k, "n=%d" % len(v), "last=%.4g" % v[-1])
PY
