"""How fast do k resident batches of the headline workload run side by side (sa_batch_start on each, then sa_batch_wait)?
The GPU-side ceiling of the pipelined loop of bench.py, without any host work in the way."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import signalalign_amd as sa
from signalalign_amd import synth
import sa_cases as cases

def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    pm = sa.Model.load(cases.MODEL_6MER)
    p = sa.default_params()
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    spec = dict(kind="gaussian", alpha=alpha, k=k, tab=tab, n_events=5000)
    sets = []
    for q in range(6):
        sets.append(sa.JobArray([synth.make_read(1000 * q + i, 5000, alpha, k, tab) for i in range(n_reads)]))
    batches = [sa.Batch(pm, p, s) for s in sets]
    for b in batches:
        b.run()
    cells = sum(b.stats().cells_forward + b.stats().cells_backward for b in batches) / len(batches)
    for kk in (1, 2, 3, 4, 6):
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            for b in batches[:kk]:
                b.start()
            for b in batches[:kk]:
                b.wait()
            best = min(best, time.perf_counter() - t0)
        print("%d side by side: %.2f ms per batch, %.3e cell updates/s" % (kk, best / kk * 1e3, cells * kk / best), flush=True)

main()
