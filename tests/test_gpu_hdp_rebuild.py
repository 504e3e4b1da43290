"""HDP rebuild, the deterministic pieces on the GPU, through the C ABI (sa_hdp_state_distr_sample, sa_hdp_finalize_distributions;
kernels in signalalign_amd/csrc/sa_hdpgrid.hip) against the CPU restatement (oracle/sa_hdp_oracle.c) and against the numbers of
the file the reference wrote."""
import numpy as np
import pytest

import signalalign_amd as sa

import hdp_cases
import sa_cases as cases

pytestmark = pytest.mark.gpu


def _oracle_sample(oracle, s):
    i = s.info
    ft = s.array("f_type")
    col = oracle.hdp_distr_sample(s.array("dp_parent"), s.array("dp_num_factor_children"), s.array("dp_depth"), s.array("observed"),
                                  s.array("gamma"), ft, s.array("f_parent"), np.where(ft == 2, -1, s.array("f_ref")),
                                  s.array("f_params"), i.mu, i.nu, 2 * i.alpha, i.beta, s.array("grid"))
    return col[s.array("observed") == 1]


def test_finalize_distributions_reproduces_the_reference_files_slopes_bit_for_bit():
    """spline_knot_slopes on the GPU, one DP per thread, the reference's elimination order: the slopes the reference stored beside
    the densities come out bit for bit (fp64 division and -ffp-contract=off: the same roundings as its x86-64 build)"""
    s = sa.HdpState(cases.NHDP)
    post, slope, grid = s.array("post"), s.array("slope"), s.array("grid")
    y, k = sa.hdp_finalize_distributions(grid, post, 1)
    assert np.array_equal(y, post) and np.array_equal(k, slope)
    # collectors of 37 samples: the division by the sample count comes first (finalize_distributions, impl/hdp.c:2551-2584)
    y37, k37 = sa.hdp_finalize_distributions(grid, post * 37.0, 37)
    np.testing.assert_allclose(y37, post, rtol=4e-16, atol=0)
    np.testing.assert_allclose(k37, slope, rtol=1e-9, atol=1e-18)
    s.close()


def test_finalize_distributions_against_the_restatement(oracle):
    rng = np.random.default_rng(3)
    grid = np.cumsum(rng.uniform(0.05, 0.4, size=257)) - 20.0          # (not equidistant: the coefficients come from the knots)
    rows = np.abs(rng.normal(size=(1000, 257))) * np.exp(-0.5 * ((grid - 5.0) / 9.0) ** 2)
    y, k = sa.hdp_finalize_distributions(grid, rows, 3)
    inv = 1.0 / 3.0
    for r in (0, 1, 499, 999):
        yr = rows[r] * inv
        assert np.array_equal(y[r], yr) and np.array_equal(k[r], oracle.hdp_spline_knot_slopes(grid, yr)), r
    with pytest.raises(sa.SaError):
        sa.hdp_finalize_distributions(grid[::-1], rows, 3)            # knots must ascend
    with pytest.raises(sa.SaError):
        sa.hdp_finalize_distributions(grid, rows, 0)


def test_one_sample_posterior_predictive_of_the_reference_files_state(oracle):
    """take_distr_sample from the state the reference's sampler stopped in (2 base factors, 1198 other factors, 352 observed DPs):
    weights on the host in the reference's order, densities and mixing on the GPU -- against the restatement.  The device's log /
    exp / pow differ from the C library's in the last bits: 1e-13 relative."""
    s = sa.HdpState(cases.NHDP)
    got = s.distr_sample()
    want = _oracle_sample(oracle, s)
    assert got.shape == want.shape == (352, 100)
    np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-300)
    # (what the file holds is the average over the run's samples, not this state's contribution: the same shape, not the same numbers)
    assert np.abs(got - s.array("post")).max() < 0.02
    s.close()


@pytest.mark.parametrize("seed,n_data,n_base", [(2, 400, 6), (9, 3000, 40)])
def test_one_sample_posterior_predictive_three_levels(oracle, tmp_path, seed, n_data, n_base):
    p = str(tmp_path / "syn.nhdp")
    hdp_cases.write_synthetic_nhdp(p, seed=seed, n_mid=6, n_leaf=9, n_data=n_data, n_base=n_base, grid=(-400.0, 420.0, 4099))
    s = sa.HdpState(p)
    got = s.distr_sample()
    # a density is exp(lgamma(a) - (log nu + 2a log beta) / 2 - log-term): one ulp of the device's log() is multiplied by the
    # factor's 2a (its data count) in the exponent -- the tolerance follows the largest factor
    two_alpha = s.array("f_params")[s.array("f_type") == 0][:, 2].max()
    np.testing.assert_allclose(got, _oracle_sample(oracle, s), rtol=4e-15 * (two_alpha + 30.0), atol=1e-300)
    dx = s.array("grid")[1] - s.array("grid")[0]
    assert np.all(np.abs(got.sum(axis=1) * dx - 1.0) < 4e-3)          # every observed DP's weights sum to one
    # the finalised spline of one sample
    y, k = sa.hdp_finalize_distributions(s.array("grid"), got, 1)
    assert np.array_equal(y, got)
    assert np.array_equal(k[0], oracle.hdp_spline_knot_slopes(s.array("grid"), got[0]))
    s.close()
