"""Stress of the register-kernel expectation pass against the memory-resident checker (no oracle: GPU against GPU), many random
shapes; prints the worst relative difference per seed.  Usage: python probes/stress_expectations.py [n_seeds]"""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import signalalign_amd as sa
import sa_cases as cases
from test_gpu_fuzz import _jobs_for

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
worst_all = 0.0
for seed in range(100, 100 + n_seeds):
    rng = np.random.default_rng(seed)
    hdp = seed % 3 == 0
    model = cases.MODEL_R73 if hdp else (cases.MODEL_6MER if seed % 2 else cases.MODEL_5MER)
    alpha, k, t10, tab, jobs = _jobs_for(model, rng, 16, False)
    pm = sa.Model.load(model, cases.NHDP if hdp else None)
    if hdp:
        pm.set_to_hdp_expected_values()
    worst = 0.0
    for expansion, trace_back, split, thr in ((50, 100, 3000 * 3000, 0.01 if not hdp else 0.05), (20, 30, 250 * 250, 0.05), (8, 12, 3000 * 3000, 0.2)):
        p = sa.default_params(threshold=thr, expansion=expansion, trace_back=trace_back, split=split)
        ft, fl, fa = sa.expect_batch(pm, p, jobs)
        gt, gl, ga = sa.expect_batch(pm, p, jobs, flags=sa.FLAG_FORCE_GENERIC)
        rel = np.abs(ft - gt) / np.maximum(np.abs(gt), 1e-6)
        worst = max(worst, float(rel.max()), float((np.abs(fl - gl) / np.maximum(np.abs(gl), 1.0)).max()))
        if hdp:
            for a, g in zip(fa, ga):
                sa_, sg = set(map(tuple, a.tolist())), set(map(tuple, g.tolist()))
                if len(sa_ ^ sg) > 2:
                    print("  seed", seed, "assignments differ:", len(sa_), len(sg), len(sa_ ^ sg))
                    worst = max(worst, 1.0)
    print("seed", seed, "hdp" if hdp else "gauss", "worst relative difference %.2e" % worst)
    worst_all = max(worst_all, worst)
print("worst over all seeds %.2e" % worst_all)
