"""Idle gaps between consecutive kernels of a rocprofv3 kernel trace (one stream): mean gap after each kernel name and the
share of the traced span that is kernel time. python tools/gap_census.py <kernel_trace.csv> [skip_first_n]"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[skip:]
gap_after = collections.defaultdict(list)
busy = 0
for a, b in zip(rows, rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    gap_after[a["Kernel_Name"].split("(")[0][-40:] + " -> " + b["Kernel_Name"].split("(")[0][-28:]].append(g)
for r in rows:
    busy += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"kernels {len(rows)}, span {span / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms ({busy / span:.1%})")
for k, v in sorted(gap_after.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print(f"{k:72s} n={len(v):5d} mean gap {sum(v) / len(v):8.1f} us  total {sum(v) / 1e3:8.2f} ms")
