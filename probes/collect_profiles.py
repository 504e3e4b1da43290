"""Copies the artefacts of probes/profile_final.sh (gpurun_out/prof_<tag>_<workload>/) into profiles/: per-workload kernel
statistics and counter summaries, and the two merged files bench.py reads (profiles/instr_mix.json, profiles/traffic.json).
Template instances of one kernel are summed under the bare kernel name.  usage: collect_profiles.py [tag] [workload ...]"""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
workloads = sys.argv[2:] or ["gaussian", "realistic", "cpg"]

def bare(name):
    name = re.sub(r"^void\s+", "", name)
    return re.sub(r"<.*$", "", name)

def merge(entries):
    """sum the *_per_step fields; fractions weighted by waves (instr) / kept from the largest (traffic)"""
    out = {}
    keys = set().union(*[e.keys() for e in entries])
    wsum = sum(e.get("waves_per_step", 1.0) for e in entries) or 1.0
    for k in keys:
        if k.endswith("_per_step"):
            out[k] = sum(e.get(k, 0.0) for e in entries)
        else:
            out[k] = sum(e.get(k, 0.0) * e.get("waves_per_step", 1.0) for e in entries) / wsum
    return out

for fname, src in (("instr_mix.json", "instr.json"), ("traffic.json", "traffic.json")):
    path = os.path.join(ROOT, "profiles", fname)
    merged = json.load(open(path)) if os.path.exists(path) else {}
    for w in workloads:
        p = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, w), src)
        if not os.path.exists(p):
            print("missing", p); continue
        d = json.load(open(p))
        groups = {}
        for k, v in d.items():
            if isinstance(v, dict):
                groups.setdefault(bare(k), []).append(v)
        merged[w] = {k: (vs[0] if len(vs) == 1 else merge(vs)) for k, vs in groups.items()}
        import subprocess, time
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        merged.setdefault("_meta", {})[w] = {"head": head, "collected": time.strftime("%Y-%m-%d %H:%M"), "tag": tag}
    # (the file's own description: which passes, which scripts -- per workload the round is in _meta)
    merged["_comment"] = ({
        "traffic.json": "HBM bytes per bench step (one sa_batch_run over the default batch of each workload, one launch per stage: "
                        "SA_GROUPS=1) from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --kernels-only "
                        "--no-secondary` (probes/profile_r04.sh, probes/traffic_from_pmc.py). Counter unit KB; FETCH_SIZE is doubled "
                        "for gfx950 as MI355X_MICROARCH.md prescribes (fetch_corrected). Template instances of a kernel are summed under "
                        "its bare name. _meta[workload] holds the commit, date and tag (round) of that workload's passes.",
        "instr_mix.json": "Wave-instructions per bench step (one sa_batch_run over the default batch of each workload, one launch per "
                          "stage) by class, from rocprofv3 --pmc SQ_INSTS_VALU/SALU/LDS and SQ_INSTS_VMEM/SMEM in two passes over "
                          "`bench.py --kernels-only --no-secondary` (probes/profile_r04.sh, probes/instr_from_pmc.py), with the wave-level "
                          "busy / wait fractions of the same passes. _meta[workload] holds the commit, date and tag (round) of the passes."}
        [fname])
    json.dump(merged, open(path, "w"), indent=1)
    print("wrote", path)
for w in workloads:
    d = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, w))
    for src, dst in (("kernel_stats.csv", "%s_%s_kernel_stats.csv" % (tag, w)), ("pmc_mix1.json", "%s_pmc_%s_mix1.json" % (tag, w)),
                     ("pmc_mix2.json", "%s_pmc_%s_mix2.json" % (tag, w)), ("traffic.json", "%s_pmc_%s_traffic_by_kernel.json" % (tag, w))):
        if os.path.exists(os.path.join(d, src)):
            shutil.copy(os.path.join(d, src), os.path.join(ROOT, "profiles", dst))

# round 4: the per-kernel times of the single-launch kernel statistics (probes/profile_r04.sh), merged into
# profiles/kernel_times.json (what bench.py's expectation leg reads for its roofline), and the bench line of the same command
import csv
kt_path = os.path.join(ROOT, "profiles", "kernel_times.json")
kt = json.load(open(kt_path)) if os.path.exists(kt_path) else {}
for w in workloads:
    d = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, w))
    cs, pf = os.path.join(d, "kernel_stats.csv"), os.path.join(d, "passes")
    if not (os.path.exists(cs) and os.path.exists(pf)):
        continue
    passes = float(open(pf).read().split()[0])
    rec = {}
    for row in csv.DictReader(open(cs)):
        k = bare(row["Name"].split("(")[0])
        r = rec.setdefault(k, {"calls": 0, "total_ms": 0.0})
        r["calls"] += int(row["Calls"])
        r["total_ms"] += float(row["TotalDurationNs"]) * 1e-6
    for k, r in rec.items():
        r["avg_ms"] = r["total_ms"] / max(r["calls"], 1)
        r["launches_per_step"] = r["calls"] / passes
        r["ms_per_step"] = r["total_ms"] / passes
    import subprocess, time
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    rec["_meta"] = {"head": head, "collected": time.strftime("%Y-%m-%d %H:%M"), "tag": tag, "kernel_passes": passes,
                    "command": "SA_GROUPS=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --workload %s --kernels-only --no-secondary "
                               "(probes/profile_r04.sh)" % w}
    kt[w] = rec
    b = os.path.join(d, "bench_kernels_only.json")
    if os.path.exists(b):
        shutil.copy(b, os.path.join(ROOT, "profiles", "bench_%s_%s_kernels_only_groups1.json" % (tag, w)))
json.dump(kt, open(kt_path, "w"), indent=1)
print("wrote", kt_path)
