"""Turns gpurun_out/prof_bench (tools/profile_bench.sh) into the tracked summaries under profiles/."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_bench")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
OUT = os.path.join(ROOT, "profiles")
os.makedirs(OUT, exist_ok=True)

shutil.copy(os.path.join(SRC, "stats", "b_kernel_stats.csv"), os.path.join(OUT, f"{tag}_bench_kernel_stats.csv"))


def per_kernel(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("void "):
            k = k[5:]                     # template instances are printed with their return type
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    return agg, n


summary = {}
for name in ("fetch", "write", "sq"):
    agg, n = per_kernel(os.path.join(SRC, name, "b_counter_collection.csv"))
    for k, v in agg.items():
        if k.startswith("rt::"):
            for c, val in v.items():
                summary.setdefault(k, {})[c] = {"sum": val, "dispatches": n[(k, c)]}
tr = summary.get("rt::k_trace", {})
launches = tr["FETCH_SIZE"]["dispatches"]
# FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  On gfx950 FETCH_SIZE counts 128-byte
# requests as 64 bytes for 16-byte-per-lane loads, hence the factor 2 on the read side.
fetch_b = tr["FETCH_SIZE"]["sum"] * 1024.0 * 2.0
write_b = tr["WRITE_SIZE"]["sum"] * 1024.0
out = {"kernel": "rt::k_trace", "launches": launches,
       "fetch_bytes_total_corrected": fetch_b, "write_bytes_total": write_b,
       "bytes_per_trace_launch": (fetch_b + write_b) / launches,
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 1 --warmup 0`; KiB -> bytes, FETCH_SIZE x2 (gfx950 counts 128-byte requests as 64 bytes)"}
json.dump(out, open(os.path.join(OUT, "hbm_traffic.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(OUT, f"{tag}_bench_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
for line in open(os.path.join(OUT, f"{tag}_bench_kernel_stats.csv")).read().splitlines()[:6]:
    print(line[:200])
sq = summary.get("rt::k_trace", {})
if "SQ_THREAD_CYCLES_VALU" in sq:
    print("k_trace avg active lanes / VALU instr:", sq["SQ_THREAD_CYCLES_VALU"]["sum"] / sq["SQ_ACTIVE_INST_VALU"]["sum"])
    print("k_trace wait share:", sq["SQ_WAIT_ANY"]["sum"] / sq["SQ_WAVE_CYCLES"]["sum"])
