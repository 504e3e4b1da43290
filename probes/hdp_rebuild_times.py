"""Times of the HDP rebuild's pieces on one GPU (DESIGN.md section 4 "HDP rebuild, round 5").  python probes/hdp_rebuild_times.py"""
import gzip, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import signalalign_amd as sa
from signalalign_amd import synth
import sa_cases as cases

H = os.path.join(cases.GOLDEN, "hdp")
data = np.array(gzip.open(os.path.join(H, "test_hdp_data.txt.gz"), "rt").read().split(), dtype=np.float64)
dps = np.array(gzip.open(os.path.join(H, "test_hdp_dps.txt.gz"), "rt").read().split(), dtype=np.int64)
s = sa.HdpState.new_tree([-1, 0, 0, 1, 1, 1, 2, 2], 3, (-10.0, 10.0, 250), (0.0, 1.0, 2.0, 10.0), gamma_alpha=[1.0, 1.0, 2.0], gamma_beta=[0.2, 0.2, 0.1])
s.pass_data(data, dps)
s.gibbs(1, 0, 1)                                   # first touch of the device
for sweeps in (1, 5):
    t = time.perf_counter()
    s.gibbs(1, sweeps * 50000 - 1, 1, seed=2)
    dt = time.perf_counter() - t
    print("test HDP (50 000 points, 8 DPs): %d x 50 000 iterations %.3f s = %.2f M iterations/s" % (sweeps, dt, sweeps * 0.05 / dt))
asg = "/tmp/asg.tsv"
open(asg, "w").write(gzip.open(os.path.join(H, "d6160b0b-a35e-43b5-947f-adaa1abade28.sm.assignments.tsv.gz"), "rt").read())
alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
nig = sa.hdp_nig_params_from_table(tab)
for layout, name, gam in ((sa.HDP_LAYOUT_FLAT, "flat", [1.0, 1.0]), (sa.HDP_LAYOUT_MULTISET, "multiset", [1.0, 1.0, 1.0])):
    for n_samples, thin in ((200, 100), (1000, 100)):
        s = sa.HdpState.new(layout, "ACGT", 6, (40.0, 140.0, 400), nig, gamma=gam)
        n = s.pass_assignment_file(asg)
        t = time.perf_counter()
        s.gibbs(n_samples, 20000, thin, seed=1)
        t1 = time.perf_counter()
        s.finalize()
        t2 = time.perf_counter()
        print("%s ACGT 6-mer, %d assignments, %d observed DPs x 400 grid points, %d factors: %d samples every %d iterations: gibbs %.3f s "
              "(%.1f ms per sample incl. %d iterations), finalize %.3f s" % (name, n, s.info.n_observed, s.info.n_factors, n_samples, thin,
                                                                         t1 - t, 1e3 * (t1 - t) / n_samples, thin, t2 - t1))
