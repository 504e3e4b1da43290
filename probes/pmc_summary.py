"""Sums rocprofv3 --pmc counters per kernel: python probes/pmc_summary.py <dir> [<dir> ...] -> JSON on stdout."""
import csv, glob, json, os, sys
from collections import defaultdict

out = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            out[k][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[k].add((f, row["Dispatch_Id"]))
res = {}
for k, c in out.items():
    c = dict(c)
    c["launches"] = len(launches[k])
    if c.get("SQ_WAVES"):
        w = c["SQ_WAVES"]
        for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM"):
            if n in c:
                c[n + "_per_wave"] = c[n] / w
    if c.get("SQ_WAVE_CYCLES"):
        for n in ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA"):
            if n in c:
                c[n + "_frac_of_wave_cycles"] = c[n] / c["SQ_WAVE_CYCLES"]
    res[k] = c
print(json.dumps(res, indent=1, sort_keys=True))
