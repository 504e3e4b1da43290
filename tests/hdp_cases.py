"""Synthetic NanoporeHDP states in the reference's .nhdp text format (serialize_nhdp / serialize_hdp, impl/nanopore_hdp.c:1077-1086,
impl/hdp.c:2919-3050), for the tests of the HDP rebuild's deterministic pieces: a three-level tree (base DP, `n_mid` middle DPs,
`n_leaf` leaves per middle DP), data in some of the leaves, a random but consistent factor tree (every data point under a factor of
its leaf, every leaf factor under a factor of the leaf's parent, every middle factor under a base factor) and base-factor parameters
computed from the data under each base factor."""
import math

import numpy as np


def _nig(mu, nu, two_alpha, beta, d):
    n = float(len(d))
    mean = float(np.mean(d))
    ssd = float(np.sum((d - mean) ** 2))
    nu_post = nu + n
    mu_post = (mu * nu + mean * n) / nu_post
    ta_post = two_alpha + n
    beta_post = beta + 0.5 * (ssd + nu * n * (mean - mu) ** 2 / nu_post)
    log_term = math.lgamma(0.5 * ta_post) - 0.5 * (math.log(nu_post) + ta_post * math.log(beta_post))
    return [mu_post, nu_post, ta_post, beta_post, log_term]


def write_synthetic_nhdp(path, seed=1, n_mid=4, n_leaf=5, n_data=400, n_base=6, grid=(-40.0, 60.0, 1500), sample_gamma=False,
                         empty_leaf_every=3):
    """returns a dict of what was written (numpy arrays) for the tests to check against"""
    rng = np.random.default_rng(seed)
    num_dps = 1 + n_mid + n_mid * n_leaf
    base_dp = 0
    dp_parent = [-1] + [0] * n_mid + [1 + m for m in range(n_mid) for _ in range(n_leaf)]
    leaves = list(range(1 + n_mid, num_dps))
    with_data = [l for j, l in enumerate(leaves) if (j % empty_leaf_every) != empty_leaf_every - 1]
    data_dp = rng.choice(with_data, size=n_data)
    centers = rng.normal(10.0, 12.0, size=n_base)
    # factor tree, built top-down then numbered in tree order
    base_of_mid = {}       # (middle dp, k) middle factors
    tree = {"children": {}, "type": {}, "ref": {}}
    nodes = []             # (type, parent node, ref)
    bases = [("b", None, base_dp) for _ in range(n_base)]
    mids = []
    for m in range(1, 1 + n_mid):
        for _ in range(int(rng.integers(1, 4))):
            mids.append(["m", int(rng.integers(0, n_base)), m])
    lfs = []
    for l in with_data:
        cands = [i for i, q in enumerate(mids) if q[2] == dp_parent[l]]
        for _ in range(int(rng.integers(1, 3))):
            lfs.append(["l", int(rng.choice(cands)), l])
    pts = []
    for i, l in enumerate(data_dp):
        cands = [j for j, q in enumerate(lfs) if q[2] == l]
        pts.append(int(rng.choice(cands)))
    # drop factors without children (the reference destroys them), bottom-up
    used_l = sorted(set(pts))
    used_m = sorted(set(lfs[j][1] for j in used_l))
    used_b = sorted(set(mids[j][1] for j in used_m))
    data = np.zeros(n_data)
    for i in range(n_data):
        b = mids[lfs[pts[i]][1]][1]
        data[i] = rng.normal(centers[b], 2.0 + 0.3 * b)
    mu, nu, alpha, beta = 8.0, 0.5, 3.0, 20.0
    lines, f_type, f_parent, f_ref, f_params = [], [], [], [], []

    def emit(t, parent, ref, params=None):
        fid = len(f_type)
        f_type.append(t); f_parent.append(parent); f_ref.append(ref); f_params.append(params or [0.0] * 5)
        if t == 0:
            lines.append("0\t-\t" + ";".join("%.17g" % v for v in params))
        else:
            lines.append("%d\t%d\t%d" % (t, parent, ref))
        return fid
    for b in used_b:
        under = [i for i in range(n_data) if mids[lfs[pts[i]][1]][1] == b]
        fb = emit(0, -1, base_dp, _nig(mu, nu, 2 * alpha, beta, data[under]))
        for m in [j for j in used_m if mids[j][1] == b]:
            fm = emit(1, fb, mids[m][2])
            for l in [j for j in used_l if lfs[j][1] == m]:
                fl = emit(1, fm, lfs[l][2])
                for i in [q for q in range(n_data) if pts[q] == l]:
                    emit(2, fl, i)
    f_type, f_parent, f_ref = np.array(f_type), np.array(f_parent), np.array(f_ref)
    nfc = np.zeros(num_dps, dtype=np.int64)
    for f in range(len(f_type)):
        if f_type[f] != 0:
            pf = f_parent[f]
            nfc[f_ref[pf]] += 1
    observed = np.zeros(num_dps, dtype=np.uint8)
    for l in data_dp:
        a = int(l)
        while a >= 0 and not observed[a]:
            observed[a] = 1
            a = dp_parent[a]
    gamma = [4.0, 1.5, 0.7]
    g0, g1, gl = grid
    with open(path, "w") as o:
        o.write("4\nACGT\n3\n")
        o.write("0\n1\n%d\n%d\n" % (1 if sample_gamma else 0, num_dps))
        o.write("\t".join("%.17g" % v for v in data) + "\n")
        o.write("\t".join(str(int(v)) for v in data_dp) + "\n")
        o.write("%.17g\t%.17g\t%.17g\t%.17g\n" % (mu, nu, alpha, beta))
        o.write("%.17g\t%.17g\t%d\n" % (g0, g1, gl))
        o.write("\t".join("%.17g" % v for v in gamma) + "\n")
        if sample_gamma:
            o.write("\t".join("%.17g" % v for v in (1.0, 1.0, 2.0)) + "\n")
            o.write("\t".join("%.17g" % v for v in (0.2, 0.2, 0.1)) + "\n")
            o.write("\t".join("%.17g" % v for v in rng.uniform(0.1, 1.0, size=num_dps)) + "\n")
            o.write("\t".join(str(int(v)) for v in rng.integers(0, 2, size=num_dps)) + "\n")
        for d in range(num_dps):
            o.write(("-" if dp_parent[d] < 0 else str(dp_parent[d])) + "\t%d\n" % nfc[d])
        for d in range(num_dps):   # collectors: zeros for observed DPs (no sample taken yet), an empty line otherwise
            o.write(("\t".join(["0"] * gl) if observed[d] else "") + "\n")
        o.write("\n".join(lines) + "\n")
    return dict(num_dps=num_dps, dp_parent=np.array(dp_parent), nfc=nfc, observed=observed, data=data, data_dp=np.array(data_dp),
                gamma=np.array(gamma), f_type=f_type, f_parent=f_parent, f_ref=f_ref, f_params=np.array(f_params), mu=mu, nu=nu,
                alpha=alpha, beta=beta, grid=grid)
