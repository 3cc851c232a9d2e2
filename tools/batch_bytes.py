#!/usr/bin/env python3
"""Per-kernel roofline of one whole-query batch (run_query_batch) from rocprofv3 output.
usage: tools/batch_bytes.py <kernel_trace.csv> <fetch counter_collection.csv> <write counter_collection.csv> [--lanes=8] [--query=K]

Durations come from the plain --kernel-trace run (counter collection serialises and slows the dispatches); FETCH_SIZE and WRITE_SIZE from their own
--pmc passes (the TCC block cannot hold both).  Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are in
KiB-sized units of 1024 B, and FETCH_SIZE reports half the bytes of wide coalesced streaming reads -> bytes_read = FETCH_SIZE * 1024 * 2,
bytes_written = WRITE_SIZE * 1024 (the same treatment as the sweep's profiles/r0N_sweep_pmc.json).  Infinity-cache hits are counted, so a kernel
that re-reads what its producer has just written shows its full algorithmic traffic here.
Output: JSON, one entry per launch of batch number K (default: the middle one) in launch order + totals per kernel class."""
import csv
import json
import sys
from collections import defaultdict

opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--"))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
trace, fetch, write = args[:3]
PEAK = 8000.0  # GB/s


def short(n):
    return n.replace("spiral::", "").replace("void ", "").split("(")[0]


def batches(rows, start_key, end_key):
    """split launch rows (sorted by start) into batches: a batch ends with the response switch"""
    out, cur = [], []
    for r in rows:
        cur.append(r)
        if "rescale" in r["name"]:
            out.append(cur)
            cur = []
    return out


def load_trace(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append({"name": short(r["Kernel_Name"]), "start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"]),
                     "grid": (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))})
    rows.sort(key=lambda r: r["start"])
    return rows


def load_counter(path, counter):
    """dispatch-ordered list of (name, grid total, value): one value per dispatch (summed over the XCD / instance rows)"""
    acc = defaultdict(float)
    meta = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        d = int(r["Dispatch_Id"])
        acc[d] += float(r["Counter_Value"])
        meta[d] = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
    return [{"name": meta[d][0], "grid_total": meta[d][1], "value": acc[d]} for d in sorted(acc)]


def split_counter(rows):
    out, cur = [], []
    for r in rows:
        cur.append(r)
        if "rescale" in r["name"]:
            out.append(cur)
            cur = []
    return out


tb = [b for b in batches(load_trace(trace), None, None) if any("sweep" in r["name"] for r in b)]
fb = [b for b in split_counter(load_counter(fetch, "FETCH_SIZE")) if any("sweep" in r["name"] for r in b)]
wb = [b for b in split_counter(load_counter(write, "WRITE_SIZE")) if any("sweep" in r["name"] for r in b)]
n_launch = len(tb[-1])
tb = [b for b in tb if len(b) == n_launch]
fb = [b for b in fb if len(b) == n_launch]
wb = [b for b in wb if len(b) == n_launch]
k = int(opts.get("query", len(tb) // 2))
T, F, W = tb[k], fb[-1], wb[-1]  # counters: the last complete batch of their runs (steady state)
assert [r["name"] for r in T] == [r["name"] for r in F] == [r["name"] for r in W], "the three runs do not hold the same launch sequence"
launches, classes = [], defaultdict(lambda: {"launches": 0, "us": 0.0, "bytes_read": 0.0, "bytes_written": 0.0})
for t, f, w in zip(T, F, W):
    us = (t["end"] - t["start"]) / 1e3
    rd, wr = f["value"] * 1024 * 2, w["value"] * 1024
    launches.append({"kernel": t["name"], "grid": "x".join(map(str, t["grid"])), "us": round(us, 1), "bytes_read": int(rd), "bytes_written": int(wr),
                     "GBps": round((rd + wr) / us / 1e3, 1), "frac_hbm": round((rd + wr) / us / 1e3 / PEAK, 3)})
    c = classes[t["name"]]
    c["launches"] += 1
    c["us"] += us
    c["bytes_read"] += rd
    c["bytes_written"] += wr
tot_us = sum(c["us"] for c in classes.values())
summary = {}
for n, c in sorted(classes.items(), key=lambda kv: -kv[1]["us"]):
    b = c["bytes_read"] + c["bytes_written"]
    summary[n] = {"launches": c["launches"], "us": round(c["us"], 1), "share": round(c["us"] / tot_us, 3), "bytes_read": int(c["bytes_read"]),
                  "bytes_written": int(c["bytes_written"]), "GBps": round(b / c["us"] / 1e3, 1), "frac_hbm": round(b / c["us"] / 1e3 / PEAK, 3),
                  "bound": "hbm" if b / c["us"] / 1e3 / PEAK >= 0.5 else "issue/latency (see DESIGN 4)"}
print(json.dumps({"lanes": int(opts.get("lanes", 8)), "batch_us": round(tot_us, 1), "launches_per_batch": n_launch, "peak_GBps": PEAK,
                  "corrections": "bytes_read = FETCH_SIZE x 1024 x 2, bytes_written = WRITE_SIZE x 1024 (MI355X_MICROARCH.md, HBM / rocprofv3)",
                  "per_kernel_class": summary, "launches": launches}, indent=1))
