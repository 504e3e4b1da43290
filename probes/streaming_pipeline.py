"""A pipeline that sees every read once, with the next batch planned while the current one is on the GPU
(sa_batch_start / sa_batch_wait): steady-state time per batch of 2000 new reads, nothing reused between batches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import signalalign_amd as sa
import sa_cases as cases
pm = sa.Model.load(cases.MODEL_6MER)
p = sa.default_params()
sets = [cases.synthetic_jobs(cases.MODEL_6MER, 2000, 5000, first_index=10000 * i) for i in range(2)]
cells = None
for mode in ("serial", "overlapped"):
    n_batches, t0, done = 12, None, 0
    prev = None
    for i in range(n_batches + 3):
        if i == 3:
            t0 = time.perf_counter()              # the first batches warm the caches
        b = sa.Batch(pm, p, sets[i % 2])
        if mode == "serial":
            b.run(); n = b.n_pairs(0); b.close()
        else:
            b.start()
            if prev is not None:
                prev.wait(); n = prev.n_pairs(0); prev.close()
            prev = b
    if prev is not None:
        prev.wait(); prev.close()
    dt = (time.perf_counter() - t0) / n_batches
    bb = sa.Batch(pm, p, sets[0]); st = bb.stats(); cells = st.cells_forward + st.cells_backward; bb.close()
    print("%-10s %.1f ms per batch of 2000 new reads = %.3g cell updates/s" % (mode, dt * 1e3, cells / dt))
