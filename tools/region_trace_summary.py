"""Summarise a rocprofv3 kernel trace of tools/region_trace.py: per region (kernels separated by > 1 ms), the span, the
per-kernel durations and the gaps."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "?")))
rows.sort()
regions, cur = [], []
for r in rows:
    if cur and r[0] - max(x[1] for x in cur) > 1_000_000:
        regions.append(cur); cur = []
    cur.append(r)
regions.append(cur)
for i, reg in enumerate(regions[-6:]):
    t0 = reg[0][0]
    span = max(x[1] for x in reg) - t0
    print(f"region -{6 - i}: {len(reg)} kernels, span {span / 1e3:.1f} us")
    if i == 5 or len(sys.argv) > 2:
        prev_end = {}
        for s, e, name, q in reg:
            print(f"   +{(s - t0) / 1e3:7.2f} .. +{(e - t0) / 1e3:7.2f}  dur {(e - s) / 1e3:6.2f}  q={q}  {name}")
