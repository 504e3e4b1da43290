"""How fast do REAL reads go?  The bundled R9.4 .npRead (10.9k template events, its own event map as the anchor source)
replicated into a batch, against the synthetic workload of the same size: cell updates/s from the device timers."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import signalalign_amd as sa          # noqa: E402
import sa_cases as cases              # noqa: E402
from oracle import sa_oracle_py as oracle   # noqa: E402 (used as a parser of the fixture only)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pm = sa.Model.load(cases.MODEL_6MER)
p = sa.default_params()
job = cases.npread_job(oracle, "r9p4_oneD.npRead", cases.MODEL_6MER)
syn = cases.synthetic_jobs(cases.MODEL_6MER, n, len(job["events"]))
for name, jobs in (("real", [job] * n), ("synthetic", syn)):
    b = sa.Batch(pm, p, jobs)
    b.run()
    t0 = time.perf_counter()
    for _ in range(3):
        b.run()
    dt = (time.perf_counter() - t0) / 3
    s = b.stats()
    cells = s.cells_forward + s.cells_backward
    print("%-9s reads %d events %d  regions %d fast %d  fwd %.2f ms bwd %.2f ms fold %.2f ms  device %.2f ms  wall %.2f ms"
          "  %.3g cell updates/s  pairs/read %d" % (name, n, len(jobs[0]["events"]), s.n_regions, s.n_fast_regions, s.ms_forward,
                                                   s.ms_backward, s.ms_fold, s.ms_total_device, dt * 1e3, cells / dt,
                                                   b.n_pairs(0)))
    b.close()
