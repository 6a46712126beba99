"""Summarise a tools/prof.sh output directory: per-kernel average duration from the
kernel trace and per-kernel average PMC counter values (per dispatch)."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.split("(")[0]
    if name.startswith("void "):
        name = name[5:]
    return name.replace("rpl::", "")


for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", f.split(out)[-1])
    for row in csv.DictReader(open(f)):
        print("  %-28s calls=%-4s avg_us=%10.1f min_us=%10.1f max_us=%10.1f pct=%s" % (
            short(row["Name"])[:28], row["Calls"], float(row["AverageNs"]) / 1e3,
            float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3, row["Percentage"]))

agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== PMC (average per dispatch)")
for k, d in agg.items():
    if not k.startswith("k_"):
        continue
    print(" ", k)
    for c, v in sorted(d.items()):
        print("    %-24s n=%-3d avg=%16.1f" % (c, len(v), sum(v) / len(v)))
