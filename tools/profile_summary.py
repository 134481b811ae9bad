"""Summaries of rocprofv3 output for the dominant kernel (the fused tower).

  python tools/profile_summary.py trace  <kernel_trace.csv>            mean launch duration and merged busy time
  python tools/profile_summary.py pmc    <fetch_counter_collection.csv> <write_counter_collection.csv>
  python tools/profile_summary.py pmc_search <fetch_counter_collection.csv> <write_counter_collection.csv>   the search / rules / tail kernels

`trace`: launches of different lanes overlap on the GPU, so next to the plain mean of (end - start) the
length of the union of the launch intervals divided by the launch count is printed; bench.py computes the
same two figures from HIP events.
`pmc`: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide
streaming read, so it is doubled. Both count traffic on the L2's memory side (Infinity Cache hits included).
"""
import collections
import csv
import json
import sys


def trace(path, key="k_tower"):
    rows = [r for r in csv.DictReader(open(path)) if key in r["Kernel_Name"]]
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
    total = sum(b - a for a, b in iv)
    busy, end = 0, -1
    for a, b in iv:
        if b > end:
            busy += b - max(a, end)
            end = b
    return {"kernel": rows[0]["Kernel_Name"].split("(")[0], "launches": len(iv),
            "avg_launch_ms_start_to_end": round(total / len(iv) / 1e6, 4),
            "avg_launch_ms_merged": round(busy / len(iv) / 1e6, 4)}


def pmc(fetch_path, write_path, key="k_tower"):
    out = {}
    for name, path, scale in (("FETCH_SIZE", fetch_path, 2.0), ("WRITE_SIZE", write_path, 1.0)):
        by_grid = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if key in r["Kernel_Name"] and r["Counter_Name"] == name:
                by_grid[int(r["Grid_Size"]) // int(r["Workgroup_Size"])].append(float(r["Counter_Value"]) * 1024 * scale)
        out[name] = {f"grid_{g}": {"dispatches": len(v), "avg_bytes": round(sum(v) / len(v))} for g, v in sorted(by_grid.items())}
        allv = [x for v in by_grid.values() for x in v]
        out[name]["all"] = {"dispatches": len(allv), "avg_bytes": round(sum(allv) / len(allv))}
    out["bytes_per_launch"] = out["FETCH_SIZE"]["all"]["avg_bytes"] + out["WRITE_SIZE"]["all"]["avg_bytes"]
    return out


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        print(json.dumps(trace(sys.argv[2]), indent=1))
    elif sys.argv[1] == "pmc_search":
        out = {}
        for key in ("k_expand_select", "k_select", "k_leaf", "k_expand_backup", "k_expand_root", "k_move_tail", "k_tails"):
            try:
                d = pmc(sys.argv[2], sys.argv[3], key="::" + key)
                out[key] = {"launches": d["FETCH_SIZE"]["all"]["dispatches"], "fetch_bytes_per_launch": d["FETCH_SIZE"]["all"]["avg_bytes"],
                            "write_bytes_per_launch": d["WRITE_SIZE"]["all"]["avg_bytes"], "bytes_per_launch": d["bytes_per_launch"]}
            except (ZeroDivisionError, KeyError):
                continue
        print(json.dumps(out, indent=1))
    else:
        print(json.dumps(pmc(sys.argv[2], sys.argv[3]), indent=1))
