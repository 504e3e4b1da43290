"""What a pipeline that sees every read once pays per batch: create, FIRST run (the pinned result buffer is sized by it),
destroy -- against the steady-state step bench.py reports."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import signalalign_amd as sa
import sa_cases as cases
pm = sa.Model.load(cases.MODEL_6MER)
p = sa.default_params()
jobs = cases.synthetic_jobs(cases.MODEL_6MER, 2000, 5000)
for rep in range(3):
    t0 = time.perf_counter(); b = sa.Batch(pm, p, jobs); t1 = time.perf_counter()
    b.run(); t2 = time.perf_counter()
    b.run(); t3 = time.perf_counter()
    n = b.n_pairs(0)
    b.close(); t4 = time.perf_counter()
    print("batch %d: create %.1f ms, first run %.1f ms, second run %.1f ms, destroy %.1f ms" %
          (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
