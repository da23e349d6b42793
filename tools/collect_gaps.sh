#!/bin/bash
# Kernel trace of the bench step (with IMT_DW_SIDE_STREAM=0: one stream, so "gap" = time with NO kernel running) and the
# idle-time summary of tools/trace_gaps.py: how much of a step sits between dependent launches.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/gaps
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export IMT_DW_SIDE_STREAM=0 IMT_ADAM_OVERLAP=0
rocprofv3 --kernel-trace --output-format csv -d $OUT/rp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/rp.err
F=$(ls $OUT/rp/*/*kernel_trace.csv | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * 0.35):int(len(rows) * 0.65)]  # timed steps only (the tail is bench.py's instrumented pass)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = []
prev_end = int(rows[0]["End_Timestamp"])
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - prev_end
    prev_end = max(prev_end, int(b["End_Timestamp"]))
    gaps.append(max(g, 0))
gaps.sort()
n = len(gaps)
print("dispatches %d, span %.2f ms, kernels busy %.2f ms (%.1f%%), idle between kernels %.2f ms" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, sum(gaps) / 1e6))
pairs = collections.Counter(); pn = collections.Counter()
prev_end = int(rows[0]["End_Timestamp"])
import re
def short(n): return re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", n)[:44]
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - prev_end
    prev_end = max(prev_end, int(b["End_Timestamp"]))
    if g > 4000:
        k = (short(a["Kernel_Name"]), short(b["Kernel_Name"])); pairs[k] += g; pn[k] += 1
steps = 15 * 0.30
for k, v in pairs.most_common(16):
    print("%7.1f us/step  n/step %4.1f  avg %6.1f us | %s -> %s" % (v / 1e3 / steps, pn[k] / steps, v / 1e3 / pn[k], k[0], k[1]))
print("gap between consecutive kernels: median %.2f us, p90 %.2f us, p99 %.2f us, mean %.2f us" % (gaps[n // 2] / 1e3, gaps[int(n * 0.9)] / 1e3, gaps[int(n * 0.99)] / 1e3, sum(gaps) / n / 1e3))
PY
tail -1 $OUT/bench.json | cut -c100-215
rm -rf $OUT/rp
