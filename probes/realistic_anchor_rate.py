"""Throughput with REALISTIC anchor density.  The synthetic workload anchors every base; a real guide alignment has an
indel every 10-50 bases and signalMachine trims 14 bases off both ends of every match run (-m 14), so only a sixth of
the bases stay anchors (tests/golden/cigars/ecoli_minus_strand.cigar: 1801 of 11333) and the band widens between them.
Here the synthetic reads keep exactly the anchors that cigar's run structure would leave."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import signalalign_amd as sa
import sa_cases as cases

toks = open(os.path.join(cases.GOLDEN, "cigars", "ecoli_minus_strand.cigar")).read().split()[10:]
runs = [(toks[i], int(toks[i + 1])) for i in range(0, len(toks), 2)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
pm = sa.Model.load(cases.MODEL_6MER)
p = sa.default_params()
dense = cases.synthetic_jobs(cases.MODEL_6MER, n, 5000)
real = []
for j, job in enumerate(dense):
    ax, ay = job["ax"], job["ay"]
    keep = np.zeros(len(ax), dtype=bool)
    pos, r = 0, (7 * j) % len(runs)            # every read starts somewhere else in the run list
    while pos < len(ax):
        op, ln = runs[r % len(runs)]; r += 1
        if op == "M":
            if ln > 28:
                keep[pos + 14: min(pos + ln - 14, len(ax))] = True
            pos += ln
        elif op == "D":
            pos += ln                           # reference bases without a read base: no anchor
    q = dict(job); q["ax"], q["ay"] = ax[keep], ay[keep]
    real.append(q)
print("anchors per read: dense %d, realistic %d" % (len(dense[0]["ax"]), len(real[0]["ax"])))
for name, jobs in (("dense", dense), ("realistic", real)):
    b = sa.Batch(pm, p, jobs); b.run()
    t0 = time.perf_counter()
    for _ in range(3): b.run()
    dt = (time.perf_counter() - t0) / 3
    s = b.stats(); cells = s.cells_forward + s.cells_backward
    print("%-9s regions %d on register kernels %d  cells/step %.3g  fwd %.2f ms bwd %.2f ms  step %.2f ms  %.3g cell updates/s  %.3g events/s"
          % (name, s.n_regions, s.n_fast_regions, cells, s.ms_forward, s.ms_backward, dt * 1e3, cells / dt, n * 5000 / dt))
    b.close()
