#!/usr/bin/env python3
"""Idle time between consecutive kernels in a rocprofv3 kernel trace (csv): where the GPU waits for the host."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the steady-state tail: last 40% of the dispatches
rows = rows[int(len(rows) * 0.6):]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = collections.Counter(); gap_n = collections.Counter()
prev_end = int(rows[0]["End_Timestamp"])
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - max(prev_end, int(a["End_Timestamp"]))
    prev_end = max(prev_end, int(b["End_Timestamp"]))
    if g > 1500:
        key = (a["Kernel_Name"][:60], b["Kernel_Name"][:60])
        gaps[key] += g; gap_n[key] += 1
print("span %.2f ms, busy %.2f ms (%.1f%%)" % (span / 1e6, busy / 1e6, 100.0 * busy / span))
for k, v in gaps.most_common(14):
    print("%8.1f us total  n=%3d  avg %6.1f us | %s -> %s" % (v / 1e3, gap_n[k], v / 1e3 / gap_n[k], k[0], k[1]))
