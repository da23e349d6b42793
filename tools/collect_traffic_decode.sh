#!/bin/bash
# HBM traffic of a decoding step by PMC counters (separate passes, counters only; corrected as MI355X_MICROARCH.md "HBM" prescribes:
# bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024): the one-launch decoder step against the launch-per-operator chain, per launch.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_decode
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BEAM_BENCH_CACHED_ONLY=1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/tools/beam_bench.py 64 128 5 > $OUT/fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/tools/beam_bench.py 64 128 5 > $OUT/write.log 2>&1
echo "write pass done"
python3 - <<PY
import csv, glob, collections, re
def load(kind):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % kind):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg
fe, wr = load("fetch"), load("write")
rows = []
for k in fe:
    n = len(fe[k]); w = wr.get(k, [0.0])
    rows.append((re.sub(r"\(anonymous namespace\)::", "", k)[:90], n, (2.0 * sum(fe[k]) / n + sum(w) / max(1, len(w))) * 1024.0))
with open("$GRAFT_REPO_ROOT/gpurun_out/pmc_decode.txt", "w") as f:
    for k, n, b in sorted(rows, key=lambda r: -r[1] * r[2])[:16]:
        line = "%-90s n=%5d  %8.2f MB/launch  %9.1f MB in all" % (k, n, b / 1e6, n * b / 1e6)
        print(line); f.write(line + "\n")
PY
