#!/bin/bash
# MFMA utilisation of the kernels of the bench step by counters (counters only, one pass): SQ_VALU_MFMA_BUSY_CYCLES (cycles a
# SIMD's matrix pipe is busy; = 32 x N for v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md cycle constants) against the wave-cycle
# and busy-cycle totals, and the instruction counts behind them.  north_star: "MFMA-utilisation counters against chip peak".
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mfma
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/t.log 2>&1
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$OUT/p/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:58]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
dur = {}
for f in glob.glob("$OUT/t/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        dur[re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:58]] = float(r["AverageNs"])
print("%-58s %5s %9s | %14s %12s | %10s %10s" % ("kernel", "n", "avg us", "MFMA busy cyc", "per SIMD-us", "insts mfma", "insts valu"))
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"])[:14]:
    m = max(1, n[k]); d = dur.get(k, 0.0)
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / m
    # 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles -> chip clock cycles of the launch = GUI / 8
    cyc = c["GRBM_GUI_ACTIVE"] / m / 8.0
    util = busy / (1024.0 * cyc) if cyc > 0 else 0.0
    print("%-58s %5d %9.1f | %14.3e %11.1f%% | %10.3e %10.3e" % (k, m, d / 1e3, busy, 100.0 * util, c["SQ_INSTS_MFMA"] / m, c["SQ_INSTS_VALU"] / m))
print("(MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch cycles), launch cycles = GRBM_GUI_ACTIVE / 8 XCDs; the bf16 peak of 2.5 PFLOP/s is 100 %)")
PY
