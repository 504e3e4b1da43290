"""Host code of the HDP rebuild's deterministic pieces: the whole .nhdp state through sa_hdp_state_load / sa_hdp_state_write
(deserialize_nhdp + deserialize_hdp, serialize_nhdp + serialize_hdp: impl/nanopore_hdp.c:1077-1115, impl/hdp.c:2868-3322).  The pin
is the file the reference itself wrote and ships (tests/golden/models/templateSingleLevelFixed.nhdp, 2 MB): it goes through
load + write BYTE FOR BYTE.  No GPU needed."""
import filecmp
import os

import numpy as np
import pytest

import signalalign_amd as sa

import hdp_cases
import sa_cases as cases


def test_reference_written_nhdp_round_trips_byte_for_byte(tmp_path):
    s = sa.HdpState(cases.NHDP)
    i = s.info
    assert (i.num_dps, i.depth, i.grid_length, i.n_data, i.n_factors, i.n_base_factors, i.n_observed, i.base_dp) == \
           (46657, 2, 100, 750, 1200, 2, 352, 46656)
    assert (i.alphabet_size, i.kmer_length, i.splines_finalized, i.has_data, i.sample_gamma) == (6, 6, 1, 1, 0)
    out = str(tmp_path / "again.nhdp")
    s.write(out)
    assert filecmp.cmp(out, cases.NHDP, shallow=False)
    # the alignment path's loader (sa_model_load) reads the same densities and slopes
    t = sa.HdpState(out)
    assert np.array_equal(t.array("post"), s.array("post")) and np.array_equal(t.array("slope"), s.array("slope"))
    s.close()
    t.close()


def test_state_arrays_are_consistent_with_the_factor_tree():
    """What the file says twice must agree: num_factor_children of every DP against the factor tree (a DP's count = the factors
    whose parent factor sits in it), observed marks against the data assignments (mark_observed_dps, impl/hdp.c:1132-1160), the
    grid against its end points (linspace)."""
    s = sa.HdpState(cases.NHDP)
    i = s.info
    ft, fp, fr = s.array("f_type"), s.array("f_parent"), s.array("f_ref")
    nfc = np.zeros(i.num_dps, dtype=np.int64)
    for f in np.nonzero(ft != 0)[0]:
        nfc[fr[fp[f]]] += 1
    assert np.array_equal(nfc, s.array("dp_num_factor_children"))
    assert np.array_equal(np.bincount(fp[fp >= 0], minlength=len(ft)), s.array("f_n_children"))
    obs = np.zeros(i.num_dps, dtype=np.uint8)
    par = s.array("dp_parent")
    for d in s.array("data_dp"):
        while d >= 0 and not obs[d]:
            obs[d] = 1
            d = par[d]
    assert np.array_equal(obs, s.array("observed")) and obs.sum() == i.n_observed
    g = s.array("grid")
    assert g[0] == i.grid_start and g[-1] == i.grid_stop and np.all(np.diff(g) > 0)
    assert sorted(set(ft[fp[ft == 2]])) == [1] and set(ft[fp[ft == 1]]) <= {0, 1}
    s.close()


@pytest.mark.parametrize("sample_gamma", [False, True])
def test_synthetic_three_level_state_round_trips(tmp_path, sample_gamma):
    p = str(tmp_path / "syn.nhdp")
    w = hdp_cases.write_synthetic_nhdp(p, seed=5, sample_gamma=sample_gamma)
    s = sa.HdpState(p)
    assert s.info.depth == 3 and s.info.num_dps == w["num_dps"] and s.info.sample_gamma == int(sample_gamma)
    assert np.array_equal(s.array("observed"), w["observed"]) and np.array_equal(s.array("f_type"), w["f_type"])
    assert np.array_equal(s.array("f_params")[w["f_type"] == 0], w["f_params"][w["f_type"] == 0])
    q = str(tmp_path / "syn2.nhdp")
    s.write(q)
    t = sa.HdpState(q)
    r = str(tmp_path / "syn3.nhdp")
    t.write(r)
    assert filecmp.cmp(q, r, shallow=False)            # idempotent ("%.17lg" round-trips every double)
    for name in ("data", "data_dp", "dp_parent", "dp_num_factor_children", "f_parent", "f_ref", "f_params", "gamma"):
        assert np.array_equal(s.array(name), t.array(name)), name
    s.close()
    t.close()


def test_malformed_states_are_refused(tmp_path):
    good = open(cases.NHDP).read().split("\n")
    p = str(tmp_path / "bad.nhdp")

    def refused(lines):
        with open(p, "w") as o:
            o.write("\n".join(lines))
        with pytest.raises(sa.SaError):
            sa.HdpState(p)
    refused(good[:9])                                   # truncated inside the header
    refused(good[:12] + good[12:2000])                  # truncated inside the DP table
    bad = list(good)
    bad[12] = "99999999\t0"                             # a parent beyond the table
    refused(bad)
    bad = list(good)
    bad[-2] = "2\t5\t99999"                             # a data index beyond the data
    refused(bad)
    bad = list(good)
    bad[-2] = "2\t0\t1"                                 # a data point under a factor of another DP
    refused(bad)
    with pytest.raises(sa.SaError):
        sa.HdpState(str(tmp_path / "missing.nhdp"))


@pytest.mark.parametrize("which", ["reference_file", "three_levels"])
def test_sample_weights_times_the_restatements_densities(oracle, tmp_path, which):
    """The host half of sa_hdp_state_distr_sample (cache_base_factor_weight / cache_prior_contribution, impl/hdp.c:2001-2044, walked
    with explicit stacks) against the CPU restatement's recursions: the CSR weights times the restatement's predictive densities,
    added in CSR order, are the restatement's collectors -- bit for bit (same sums in the same order).  No GPU."""
    if which == "reference_file":
        s = sa.HdpState(cases.NHDP)
    else:
        p = str(tmp_path / "syn.nhdp")
        hdp_cases.write_synthetic_nhdp(p, seed=21, n_mid=5, n_leaf=7, n_data=900, n_base=12)
        s = sa.HdpState(p)
    i = s.info
    ft, fpar, grid = s.array("f_type"), s.array("f_params"), s.array("grid")
    rs, col, w = s.sample_weights()
    assert rs[0] == 0 and rs[-1] == len(col) == len(w) and np.all(np.diff(rs) >= 1)
    for r in range(i.n_observed):
        assert np.all(np.diff(col[rs[r]:rs[r + 1]]) > 0)          # a row's columns ascend: the order the reference adds them in
        assert abs(w[rs[r]:rs[r + 1]].sum() - 1.0) < 1e-12        # ... and its weights sum to one
    pdf = [oracle.hdp_posterior_predictive(fpar[F], grid) for F in np.nonzero(ft == 0)[0]]
    pdf.append(oracle.hdp_prior_predictive(i.mu, i.nu, 2 * i.alpha, i.beta, grid))
    want = oracle.hdp_distr_sample(s.array("dp_parent"), s.array("dp_num_factor_children"), s.array("dp_depth"), s.array("observed"),
                                   s.array("gamma"), ft, s.array("f_parent"), np.where(ft == 2, -1, s.array("f_ref")), fpar,
                                   i.mu, i.nu, 2 * i.alpha, i.beta, grid)[s.array("observed") == 1]
    for r in range(i.n_observed):
        acc = np.zeros(i.grid_length)
        for e in range(rs[r], rs[r + 1]):
            acc = acc + w[e] * pdf[col[e]]
        assert np.array_equal(acc, want[r]), r
    s.close()


def test_public_header_is_self_contained(tmp_path):
    """include/signalalign_hip.h compiles on its own as C11 and as C++ (round 4: it used size_t without <stddef.h>)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name, cc, std in (("t.c", "gcc", "-std=c11"), ("t.cpp", "g++", "-std=c++17")):
        src = tmp_path / name
        src.write_text('#include "signalalign_hip.h"\nint main(void) { sa_pair16_t r = sa_pair16_pack(5, 1, 2, 0, 3); '
                       'return (int) sa_pair16_unpack(r).x - 1; }\n')
        subprocess.run([cc, std, "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-o", str(tmp_path / "t.out"), str(src)],
                       check=True)
        assert subprocess.run([str(tmp_path / "t.out")]).returncode == 0
