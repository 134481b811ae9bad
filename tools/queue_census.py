"""Which hardware queues the kernels of a rocprofv3 kernel trace went through, per 50 ms of the trace:
python tools/queue_census.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
bins = collections.defaultdict(collections.Counter)
for r in rows:
    bins[(int(r["Start_Timestamp"]) - t0) // 50_000_000][r.get("Queue_Id", "?")] += 1
for b in sorted(bins):
    print(f"{b * 50:6d} ms: " + "  ".join(f"queue {q}: {n}" for q, n in sorted(bins[b].items())))
