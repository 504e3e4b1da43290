"""HBM bytes per bench step of every kernel, from two rocprofv3 --pmc passes over `bench.py --kernels-only`
(FETCH_SIZE in one, WRITE_SIZE in the other: they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").
Counter unit: KB.  gfx950 correction (same guide, HBM section): FETCH_SIZE reports half of the bytes of a coalesced
streaming read -> doubled; the guide calibrates that for 16 B/lane loads, ours are 8 B/lane (uncalibrated: read the
fetch side as +-2x).  Usage: traffic_from_pmc.py <fetch_dir> <write_dir> <kernel_passes>  -> JSON on stdout."""
import csv, glob, json, os, sys
from collections import defaultdict

fetch_dir, write_dir, passes = sys.argv[1], sys.argv[2], float(sys.argv[3])


def sums(d, counter):
    out, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            out[k] += float(row["Counter_Value"]) * 1024.0
            n[k] += 1
    return out, n


fe, nf = sums(fetch_dir, "FETCH_SIZE")
wr, nw = sums(write_dir, "WRITE_SIZE")
res = {}
for k in sorted(set(fe) | set(wr)):
    f_raw, w = fe.get(k, 0.0) / passes, wr.get(k, 0.0) / passes
    res[k] = {"bytes_per_step": 2.0 * f_raw + w, "fetch_raw_bytes_per_step": f_raw, "fetch_corrected_bytes_per_step": 2.0 * f_raw,
              "write_bytes_per_step": w, "launches_per_step": nf.get(k, nw.get(k, 0)) / passes}
print(json.dumps(res, indent=1))
