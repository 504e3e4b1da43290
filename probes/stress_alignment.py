"""Stress of the default kernels (register / strip / ring, HDP emission plane, device planner, packed result records) against
the reference-ordered kernels (SA_FLAG_EXACT) on many random shapes: GPU against GPU, the tests' 1e-5 bar and row order.
Usage: python probes/stress_alignment.py [n_seeds]"""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import signalalign_amd as sa
import sa_cases as cases
from test_gpu_fuzz import _jobs_for

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
worst_all, only_all = 0, 0
for seed in range(200, 200 + n_seeds):
    rng = np.random.default_rng(seed)
    kind = seed % 4
    model = [cases.MODEL_6MER, cases.MODEL_5MER, cases.MODEL_CPG, cases.MODEL_R73][kind]
    alpha, k, t10, tab, jobs = _jobs_for(model, rng, 20, kind == 2)
    pm = sa.Model.load(model, cases.NHDP if kind == 3 else None)
    if kind == 3:
        pm.set_to_hdp_expected_values()
    amb = sa.default_ambig({"X": "CE"}) if kind == 2 else None
    worst, only = 0, 0
    for expansion, trace_back, split, thr in ((50, 100, 3000 * 3000, 0.01 if kind != 3 else 0.05), (20, 30, 250 * 250, 0.1), (8, 12, 3000 * 3000, 0.3)):
        p = sa.default_params(threshold=thr, expansion=expansion, trace_back=trace_back, split=split)
        a = sa.Batch(pm, p, jobs, ambig=amb)
        a.run()
        e = sa.Batch(pm, p, jobs, ambig=amb, flags=sa.FLAG_EXACT)
        e.run()
        for j in range(len(jobs)):
            w, n1 = cases.compare_pairs(a.pairs(j), e.pairs(j), 100, thr)
            assert cases.same_order(a.pairs(j), e.pairs(j)), (seed, j)
            worst, only = max(worst, w), only + n1
        a.close()
        e.close()
    print("seed", seed, ["6mer", "5mer", "cpg", "hdp"][kind], "worst |dp| %d e-7, rows on one side only %d" % (worst, only))
    worst_all, only_all = max(worst_all, worst), only_all + only
print("worst over all seeds %d e-7; rows near the threshold on one side only: %d" % (worst_all, only_all))
