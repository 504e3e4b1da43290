"""Wave-instructions per bench step of every kernel, from the two instruction-mix counter passes of probes/profile_final.sh
(SQ_INSTS_VALU / SALU / LDS in one, SQ_INSTS_VMEM / SMEM in the other, both over `bench.py --kernels-only`).
Usage: instr_from_pmc.py <mix1.json> <mix2.json> <kernel_passes>  -> JSON on stdout (per kernel: instructions per step by
class, waves per step, and the wave-level fractions the SQ counters give)."""
import json, sys

m1, m2, passes = json.load(open(sys.argv[1])), json.load(open(sys.argv[2])), float(sys.argv[3])
res = {}
for k in sorted(set(m1) | set(m2)):
    a, b = m1.get(k, {}), m2.get(k, {})
    r = {"waves_per_step": a.get("SQ_WAVES", b.get("SQ_WAVES", 0.0)) / passes}
    for name, src in (("valu", a), ("salu", a), ("lds", a), ("vmem", b), ("smem", b)):
        key = "SQ_INSTS_" + name.upper()
        if key in src:
            r[name + "_per_step"] = src[key] / passes
    r["instructions_per_step"] = sum(v for n, v in r.items() if n.endswith("_per_step") and n != "waves_per_step")
    for key in ("SQ_ACTIVE_INST_VALU_frac_of_wave_cycles", "SQ_WAIT_INST_ANY_frac_of_wave_cycles"):
        if key in a:
            r[key] = a[key]
    for key in ("SQ_WAIT_ANY_frac_of_wave_cycles", "SQ_ACTIVE_INST_ANY_frac_of_wave_cycles"):
        if key in b:
            r[key] = b[key]
    res[k] = r
print(json.dumps(res, indent=1))
