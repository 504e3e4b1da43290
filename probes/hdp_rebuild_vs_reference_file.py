"""The one statistical pin the reference offers for its Gibbs sampler: templateSingleLevelFixed.nhdp holds the DATA it was built from
(750 events, their leaf DPs), its hyperparameters and the densities the reference's own sampler averaged.  The same data through this
library's sampler (flat ACEGOT 6-mer model, the file's grid, base distribution and concentration parameters) must give the same
posterior predictive densities up to Monte-Carlo noise.  python probes/hdp_rebuild_vs_reference_file.py [n_samples] [seed...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import signalalign_amd as sa
import sa_cases as cases

n_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seeds = [int(a) for a in sys.argv[2:]] or [1, 2]
ref = sa.HdpState(cases.NHDP)
i = ref.info
grid, post_ref, rows_ref = ref.array("grid"), ref.array("post"), ref.array("row_of_dp")
data, data_dp = ref.array("data"), ref.array("data_dp")
obs = np.flatnonzero(ref.array("observed"))
dxs = np.diff(grid)
results = []
for seed in seeds:
    s = sa.HdpState.new(sa.HDP_LAYOUT_FLAT, "ACEGOT", 6, (float(grid[0]), float(grid[-1]), len(grid)), (i.mu, i.nu, i.alpha, i.beta),
                        gamma=[float(g) for g in ref.array("gamma")])
    s.pass_data(data, data_dp)
    s.gibbs(n_samples, 40 * len(data), 4 * len(data), seed=seed)
    s.finalize()
    post, rows = s.array("post"), s.array("row_of_dp")
    assert np.array_equal(np.flatnonzero(s.array("observed")), obs)
    l1, peak = [], []
    for d in obs:
        a, b = post[rows[d]], post_ref[rows_ref[d]]
        l1.append(float(np.sum(0.5 * (np.abs(a - b)[1:] + np.abs(a - b)[:-1]) * dxs)))
        peak.append(float(np.abs(a - b).max() / b.max()))
    l1, peak = np.array(l1), np.array(peak)
    n_of = np.array([(data_dp == d).sum() for d in obs])
    print("seed %d: %d samples; L1 distance to the reference's stored density per observed DP: median %.4f, 95 %% %.4f, worst %.4f (DP %d, %d data points); "
          "worst |difference| / peak: median %.4f, worst %.4f; base DP: L1 %.4f" % (seed, n_samples, np.median(l1), np.quantile(l1, 0.95), l1.max(),
          obs[int(l1.argmax())], n_of[int(l1.argmax())], np.median(peak), peak.max(), l1[list(obs).index(i.base_dp)]), flush=True)
    results.append(post[[rows[d] for d in obs]])
if len(results) > 1:   # the Monte-Carlo noise itself: two seeds of this sampler against each other
    a, b = results[0], results[1]
    l1 = np.sum(0.5 * (np.abs(a - b)[:, 1:] + np.abs(a - b)[:, :-1]) * dxs, axis=1)
    print("seed %d against seed %d: median %.4f, 95 %% %.4f, worst %.4f" % (seeds[0], seeds[1], np.median(l1), np.quantile(l1, 0.95), l1.max()))
