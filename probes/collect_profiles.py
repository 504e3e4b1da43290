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
    json.dump(merged, open(path, "w"), indent=1)
    print("wrote", path)
for w in workloads:
    d = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, w))
    for src, dst in (("kernel_stats.csv", "%s_%s_kernel_stats.csv" % (tag, w)), ("pmc_mix1.json", "%s_pmc_%s_mix1.json" % (tag, w)),
                     ("pmc_mix2.json", "%s_pmc_%s_mix2.json" % (tag, w)), ("traffic.json", "%s_pmc_%s_traffic_by_kernel.json" % (tag, w))):
        if os.path.exists(os.path.join(d, src)):
            shutil.copy(os.path.join(d, src), os.path.join(ROOT, "profiles", dst))
