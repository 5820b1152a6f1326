"""Sums rocprofv3 counter rows per kernel and prints VALU figures (development aid)."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if "rt::" not in k:
        continue
    line = k
    if "SQ_ACTIVE_INST_VALU" in v and v.get("SQ_BUSY_CYCLES"):
        line += "  valu_issue_share %.3f" % (v["SQ_ACTIVE_INST_VALU"] * 4 / (v["SQ_BUSY_CYCLES"] * 32))
        line += "  lanes/valu_inst %.1f" % (v["SQ_THREAD_CYCLES_VALU"] / v["SQ_ACTIVE_INST_VALU"])
        line += "  wait_any/wave_cycles %.3f" % (v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"])
        line += "  wait_inst/wave_cycles %.3f" % (v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"])
        line += "  valu_insts %.4g  busy_cycles %.4g" % (v["SQ_INSTS_VALU"], v["SQ_BUSY_CYCLES"])
    else:
        line += "  " + str(dict(v))
    print(line)
