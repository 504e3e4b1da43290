"""The CPU restatement of the HDP rebuild's deterministic pieces (oracle/sa_hdp_oracle.c) against numbers the REFERENCE wrote:
tests/golden/models/templateSingleLevelFixed.nhdp stores, for each of its 352 observed DPs, the density on the sampling grid AND the
spline slopes the reference computed from it, and for each base factor the normal-inverse-gamma parameters its sampler cached.
"%.17lg" round-trips doubles, so these are exact vectors."""
import numpy as np

import signalalign_amd as sa

import hdp_cases
import sa_cases as cases


def _state_arrays(s):
    return {n: s.array(n) for n in ("grid", "post", "slope", "f_type", "f_parent", "f_ref", "f_params", "data", "data_dp", "dp_parent",
                                    "dp_num_factor_children", "dp_depth", "observed", "gamma", "row_of_dp")}


def test_spline_slopes_of_the_reference_file_bit_for_bit(oracle):
    s = sa.HdpState(cases.NHDP)
    a = _state_arrays(s)
    assert np.array_equal(a["grid"], oracle.hdp_linspace(s.info.grid_start, s.info.grid_stop, s.info.grid_length))
    assert a["post"].shape == (352, 100)
    for r in range(a["post"].shape[0]):
        assert np.array_equal(oracle.hdp_spline_knot_slopes(a["grid"], a["post"][r]), a["slope"][r]), r
    s.close()


def test_base_factor_parameters_of_the_reference_file(oracle):
    """cached incrementally by the sampler (add / remove_update_base_factor_params, impl/hdp.c:424-468); the batch posterior over
    the data under each base factor agrees to rounding"""
    s = sa.HdpState(cases.NHDP)
    a, i = _state_arrays(s), s.info
    base = np.arange(len(a["f_type"]))
    for f in range(len(base)):
        if a["f_parent"][f] >= 0:
            base[f] = base[a["f_parent"][f]]
    seen = 0
    for F in np.nonzero(a["f_type"] == 0)[0]:
        d = a["data"][a["f_ref"][(a["f_type"] == 2) & (base == F)]]
        p5 = oracle.hdp_nig_posterior(i.mu, i.nu, 2 * i.alpha, i.beta, d)
        np.testing.assert_allclose(p5, a["f_params"][F], rtol=1e-12, atol=0)
        seen += len(d)
    assert seen == i.n_data
    s.close()


def test_one_sample_posterior_predictive_properties(oracle, tmp_path):
    """take_distr_sample is parity unpinned (the file's densities are averages over a Gibbs run): properties.  Every observed DP's
    collector integrates to one over a grid that covers it (its weights sum to one), unobserved DPs collect nothing, and the base
    DP's mixture is its factors' weights (children / (gamma + n)) times their predictive densities plus the prior's share."""
    p = str(tmp_path / "syn.nhdp")
    w = hdp_cases.write_synthetic_nhdp(p, seed=11, grid=(-400.0, 420.0, 8000))
    s = sa.HdpState(p)
    a, i = _state_arrays(s), s.info
    f_dp = np.where(a["f_type"] == 2, -1, a["f_ref"])
    col = oracle.hdp_distr_sample(a["dp_parent"], a["dp_num_factor_children"], a["dp_depth"], a["observed"], a["gamma"], a["f_type"],
                                  a["f_parent"], f_dp, a["f_params"], i.mu, i.nu, 2 * i.alpha, i.beta, a["grid"])
    dx = a["grid"][1] - a["grid"][0]
    integ = col.sum(axis=1) * dx
    assert np.all(np.abs(integ[a["observed"] == 1] - 1.0) < 2e-3), (integ[a["observed"] == 1].min(), integ[a["observed"] == 1].max())
    assert np.all(col[a["observed"] == 0] == 0.0) and (a["observed"] == 0).sum() > 0
    g0, n0 = a["gamma"][0], a["dp_num_factor_children"][i.base_dp]
    want = np.zeros(i.grid_length)
    for F in np.nonzero(a["f_type"] == 0)[0]:
        want += (np.sum(a["f_parent"] == F) / (g0 + n0)) * oracle.hdp_posterior_predictive(a["f_params"][F], a["grid"])
    want += (g0 / (g0 + n0)) * oracle.hdp_prior_predictive(i.mu, i.nu, 2 * i.alpha, i.beta, a["grid"])
    np.testing.assert_allclose(col[i.base_dp], want, rtol=1e-13, atol=0)
    s.close()
