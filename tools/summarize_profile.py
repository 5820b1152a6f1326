"""Turns gpurun_out/prof_<workload> (tools/profile_bench.sh) into the tracked summaries under
profiles/:  <tag>_<workload>_kernel_stats.csv, <tag>_<workload>_pmc_summary.json and
pmc_<workload>.json, the per-launch figures that bench.py replays into its roofline entries.

    python tools/summarize_profile.py r02 bunny
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
workload = sys.argv[2] if len(sys.argv) > 2 else "bunny"
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + workload)
OUT = os.path.join(ROOT, "profiles")
os.makedirs(OUT, exist_ok=True)

shutil.copy(os.path.join(SRC, "stats", "b_kernel_stats.csv"), os.path.join(OUT, f"{tag}_{workload}_kernel_stats.csv"))


def per_kernel(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("void "):
            k = k[5:]                     # template instances are printed with their return type
        k = k.split("<")[0]
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


def figures(*kernels):
    """Per-launch HBM bytes and VALU figures of one kernel, or of several taken together.  FETCH_SIZE / WRITE_SIZE are reported in
    KiB per dispatch; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes for wide loads,
    hence the factor 2 on the read side.  SQ_BUSY_CYCLES is summed over the 32 shader engines, each
    with 32 SIMDs (256 CUs x 4 / 32), and SQ_ACTIVE_INST_VALU counts quad-cycles per SIMD."""
    d = {}
    for k in kernels:
        for c, v in summary.get(k, {}).items():
            e = d.setdefault(c, {"sum": 0.0, "dispatches": 0})
            e["sum"] += v["sum"]; e["dispatches"] += v["dispatches"]
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
        return {}
    launches = d["FETCH_SIZE"]["dispatches"]
    fetch_b = d["FETCH_SIZE"]["sum"] * 1024.0 * 2.0
    write_b = d["WRITE_SIZE"]["sum"] * 1024.0
    out = {"launches": launches, "fetch_bytes_total_corrected": fetch_b, "write_bytes_total": write_b,
           "hbm_bytes_per_launch": (fetch_b + write_b) / launches}
    if "SQ_THREAD_CYCLES_VALU" in d:
        out["active_lanes_per_valu_inst"] = d["SQ_THREAD_CYCLES_VALU"]["sum"] / d["SQ_ACTIVE_INST_VALU"]["sum"]
        out["valu_issue_share"] = d["SQ_ACTIVE_INST_VALU"]["sum"] * 4.0 / (d["SQ_BUSY_CYCLES"]["sum"] * 32.0)
        out["wait_share_of_wave_cycles"] = d["SQ_WAIT_ANY"]["sum"] / d["SQ_WAVE_CYCLES"]["sum"]
    return out


pmc = {"collected": f"{tag}, `bench.py --workload {workload} --steps 1 --warmup 0 --no-cpu-baseline`, FETCH_SIZE / WRITE_SIZE / SQ counters in separate passes",
       "trace": figures("rt::k_trace", "rt::k_trace_shadow", "rt::k_sweep", "rt::k_sweep_shadow", "rt::k_trace_retry", "rt::k_trace_shadow_retry"),
       "shade": figures("rt::k_shade"),
       "k_trace": figures("rt::k_trace"), "k_trace_shadow": figures("rt::k_trace_shadow"),
       "k_sweep": figures("rt::k_sweep"), "k_sweep_shadow": figures("rt::k_sweep_shadow"),
       "k_trace_retry": figures("rt::k_trace_retry", "rt::k_trace_shadow_retry")}
json.dump(pmc, open(os.path.join(OUT, f"pmc_{workload}.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(OUT, f"{tag}_{workload}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(pmc, indent=1))
for line in open(os.path.join(OUT, f"{tag}_{workload}_kernel_stats.csv")).read().splitlines()[:7]:
    print(line[:200])
