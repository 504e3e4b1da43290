"""HDP rebuild, the host side without the sweeps (which need the GPU for their distribution samples: tests/test_gpu_hdp_rebuild.py):
model construction, the normal-inverse-gamma parameters of a lookup table, data passing and the initial factor tree, serialisation.

Mirrors of the reference's tests with their literal numbers (paths relative to the upstream tree):
  tests/hdpTests.c:40-70          test_mle_params            (no candidate within a factor 4 of the estimate has a higher likelihood)
  tests/nanoporeHdpTests.c:18-100 the word / multiset arithmetic behind the tree layouts
  tests/nanoporeHdpTests.c:272-345 test_serialization, first two stages (structure only, with data): write -> load -> write, same bytes
Fixtures: tests/golden/hdp/ holds the reference's own test data as data (tests/test_hdp/data.txt, dps.txt;
tests/test_alignments/simple_alignment.tsv; tests/test_assignment_files/d6160b0b-...sm.assignments.tsv), gzip-compressed.
"""
import gzip
import itertools
import math
import os

import numpy as np
import pytest

import signalalign_amd as sa
from signalalign_amd import synth

import sa_cases as cases

HDP = os.path.join(cases.GOLDEN, "hdp")
NIG = (56.7, 77.4, 19.8, 64665.9)


def test_mle_params():
    mus = np.array([-20.1, 2.8, -11.7, -39.3, -0.4])
    taus = np.array([0.01, 0.005, 0.0023, 0.013, 0.008])
    tab = np.zeros((5, 5))
    tab[:, 0], tab[:, 1] = mus, 1.0 / np.sqrt(taus)              # (the table holds level sds: precision = 1 / sd^2)
    mu0, nu, alpha, beta = sa.hdp_nig_params_from_table(tab)

    def ll(m0, n, a, b):   # norm_gamma_joint_log_likelihood (tests/hdpTests.c:23-38)
        dens = (b ** a / math.gamma(a)) * taus ** (a - 1.0) * np.exp(-b * taus) * np.sqrt(n * taus / (2 * math.pi)) * \
            np.exp(-(n * taus / 2.0) * (mus - m0) ** 2)
        return float(np.log(dens).sum())
    best = ll(mu0, nu, alpha, beta)
    for i, j, k, l in itertools.product(range(-2, 3), repeat=4):
        assert ll(2.0 ** i * mu0, 2.0 ** j * nu, 2.0 ** k * alpha, 2.0 ** l * beta) <= best + .0000001
    # the two special functions behind the Newton iteration, against scipy's
    from scipy.special import digamma, polygamma
    for x in (0.05, 0.3, 1.0, 2.5, 7.7, 13.2, 100.0, 1e4):
        assert abs(sa.lib().sa_hdp_digamma(x) - digamma(x)) <= 4e-15 * max(1.0, abs(digamma(x)))
        assert abs(sa.lib().sa_hdp_trigamma(x) - polygamma(1, x)) <= 4e-15 * max(1.0, polygamma(1, x))
    # a real lookup table (the R7.3 template model): the estimate is a stationary point of the same likelihood
    alpha_, k_, t10, table = synth.parse_model_table(cases.MODEL_R73)
    m0, n0, a0, b0 = sa.hdp_nig_params_from_table(table)
    assert 55 < m0 < 65 and n0 > 0 and a0 > 1 and b0 > 0


def _multiset_number(n, k):
    return math.comb(n + k - 1, k)


def test_tree_layouts():
    # flat_hdp_num_dps / multiset_hdp_num_dps / middle_2_nts_hdp_num_dps / purine_composition_hdp_num_dps / group_multiset_hdp_num_dps
    # (impl/nanopore_hdp.c:489-1060) and the parent rule of every layout on k-mers whose parent can be said by hand
    grid = (0.0, 100.0, 100)
    flat = sa.HdpState.new(sa.HDP_LAYOUT_FLAT, "ACGT", 6, grid, NIG, gamma=[4.0, 20.0])
    assert flat.info.num_dps == 4 ** 6 + 1 and flat.info.depth == 2 and flat.info.base_dp == 4 ** 6
    assert np.all(flat.array("dp_parent")[:-1] == 4 ** 6) and flat.array("dp_parent")[-1] == -1
    assert flat.kmer_dp("AAAAAC") == 1 and flat.kmer_dp("AACAAA") == 64 and flat.kmer_dp("AANAAA") == -1
    ms = sa.HdpState.new(sa.HDP_LAYOUT_MULTISET, "ACEGOT", 4, grid, NIG, gamma=[1.0, 1.0, 1.0])
    L = 6 ** 4
    assert ms.info.num_dps == L + _multiset_number(6, 4) + 1 and ms.info.depth == 3
    pa = ms.array("dp_parent")
    assert pa[ms.kmer_dp("ACGT")] == pa[ms.kmer_dp("TGCA")] == pa[ms.kmer_dp("GATC")]          # one multiset, one parent
    assert pa[ms.kmer_dp("AAAA")] == L and pa[ms.kmer_dp("TTTT")] == L + _multiset_number(6, 4) - 1   # first and last multiset
    assert pa[ms.kmer_dp("AACC")] != pa[ms.kmer_dp("AAAC")]
    assert len(set(pa[:L].tolist())) == _multiset_number(6, 4) and np.all(pa[L:-1] == ms.info.base_dp)
    mid = sa.HdpState.new(sa.HDP_LAYOUT_MIDDLE_NTS, "ACGT", 6, grid, NIG, gamma_alpha=[1, 1, 1], gamma_beta=[.2, .2, .2])
    assert mid.info.num_dps == 4 ** 6 + 16 + 1 and mid.info.sample_gamma == 1
    pa = mid.array("dp_parent")
    assert pa[mid.kmer_dp("AACGTT")] == pa[mid.kmer_dp("TTCGAA")] == 4 ** 6 + 4 * 1 + 2          # middle letters C, G
    np.testing.assert_allclose(mid.array("gamma"), [5.0, 5.0, 5.0])                               # alpha / beta: the prior's mean
    comp = sa.HdpState.new(sa.HDP_LAYOUT_COMPOSITION, "AGCEOT", 4, grid, NIG, gamma=[1, 1, 1], groups=[1, 1, 0, 0, 0, 0])
    assert comp.info.num_dps == 6 ** 4 + 5 + 1 and sa.lib().sa_hdp_state_kmer_dp(comp._h, b"AAAA") == 0   # (sorted alphabet ACEGOT)
    pa = comp.array("dp_parent")
    assert pa[comp.kmer_dp("CCTT")] == 6 ** 4 and pa[comp.kmer_dp("ACTG")] == 6 ** 4 + 2 and pa[comp.kmer_dp("GAGA")] == 6 ** 4 + 4
    grp = sa.HdpState.new(sa.HDP_LAYOUT_GROUP_MULTISET, "ACEGOT", 3, grid, NIG, gamma=[1, 1, 1], groups=[0, 1, 1, 2, 1, 3])
    assert grp.info.num_dps == 6 ** 3 + _multiset_number(4, 3) + 1
    pa = grp.array("dp_parent")
    assert pa[grp.kmer_dp("ACG")] == pa[grp.kmer_dp("AEG")] == pa[grp.kmer_dp("GOA")] != pa[grp.kmer_dp("ACT")]
    for bad in (dict(gamma=[1.0]), dict(gamma=[1.0, -2.0, 1.0]), dict()):
        with pytest.raises(sa.SaError):
            sa.HdpState.new(sa.HDP_LAYOUT_MULTISET, "ACGT", 5, grid, NIG, **bad)
    with pytest.raises(sa.SaError):
        sa.HdpState.new(sa.HDP_LAYOUT_FLAT, "ACGA", 5, grid, NIG, gamma=[1, 1])                    # letters must be distinct
    with pytest.raises(sa.SaError):
        sa.HdpState.new_tree([-1, 0, 0, 1], 3, grid, NIG, gamma=[1, 1, 1])                         # a leaf (2) above the leaf depth
    with pytest.raises(sa.SaError):
        sa.HdpState.new_tree([-1, -1, 0, 1], 2, grid, NIG, gamma=[1, 1])                           # two roots


def _reference_hdp_test_data():
    data = np.array(gzip.open(os.path.join(HDP, "test_hdp_data.txt.gz"), "rt").read().split(), dtype=np.float64)
    dps = np.array(gzip.open(os.path.join(HDP, "test_hdp_dps.txt.gz"), "rt").read().split(), dtype=np.int64)
    keep = dps != 4                                   # (tests/nanoporeHdpTests.c:292: DP 4 stays unobserved)
    return data[keep], dps[keep]


def _reference_hdp():
    return sa.HdpState.new_tree([-1, 0, 0, 1, 1, 1, 2, 2], 3, (-10.0, 10.0, 250), (0.0, 1.0, 2.0, 10.0),
                                gamma_alpha=[1.0, 1.0, 2.0], gamma_beta=[0.2, 0.2, 0.1])


def test_data_passing_and_serialisation_of_the_reference_test_hdp(tmp_path, oracle):
    s = _reference_hdp()
    a, b = str(tmp_path / "a.hdp"), str(tmp_path / "b.hdp")
    s.write(a)                                        # structure only
    sa.HdpState(a).write(b)
    assert open(a).read() == open(b).read()
    data, dps = _reference_hdp_test_data()
    assert len(data) == 50000 - 10123 and set(dps.tolist()) == {3, 5, 6, 7}
    s.pass_data(data, dps)
    i = s.info
    assert (i.has_data, i.splines_finalized, i.n_data, i.n_base_factors) == (1, 0, len(data), 1)
    obs = s.array("observed")
    assert obs.tolist() == [1, 1, 1, 1, 0, 1, 1, 1] and i.n_observed == 7
    # init_factors (impl/hdp.c:1440-1547): one base factor, one factor per observed DP below it, one per data point
    ft, fp, fr, fn = s.array("f_type"), s.array("f_parent"), s.array("f_ref"), s.array("f_n_children")
    assert i.n_factors == 1 + 6 + len(data) and (ft == 0).sum() == 1 and (ft == 1).sum() == 6
    assert fn[0] == 2 and sorted(fr[ft == 1].tolist()) == [1, 2, 3, 5, 6, 7]
    nc = s.array("dp_num_factor_children")
    assert nc[0] == 2 and nc[1] == 2 and nc[2] == 2 and nc[4] == 0 and nc[3] == (dps == 3).sum() and nc[7] == (dps == 7).sum()
    # the base factor's parameters: the posterior of the normal-inverse-gamma prior given ALL the data at once
    want = oracle.hdp_nig_posterior(0.0, 1.0, 4.0, 10.0, data)
    np.testing.assert_allclose(s.array("f_params")[0], want, rtol=1e-12)
    s.write(a)                                        # with data
    c = sa.HdpState(a)
    c.write(b)
    assert open(a).read() == open(b).read()
    assert np.array_equal(c.array("data"), data) and np.array_equal(c.array("data_dp"), dps)
    # new data replace the old ones (reset_hdp_data + pass_data_to_hdp), a data point in an inner DP is refused, and sampling
    # needs the GPU-side collectors: without a device it fails loudly
    with pytest.raises(sa.SaError):
        s.pass_data([1.0, 2.0], [1, 3])
    s.pass_data(data[:100], dps[:100])
    assert s.info.n_data == 100 and s.info.n_factors == 1 + len(set(dps[:100].tolist())) + len({1, 2} & {1 if d in (3, 5) else 2 for d in dps[:100]}) + 100
    if sa.device_count() == 0 and not os.environ.get("SA_SAMPLER_STUB"):   # (probes/host_asan.sh stubs the GPU sampler out)
        with pytest.raises(sa.SaError) as ei:
            s.gibbs(10, 10, 10)
        assert ei.value.code == -3


def test_assignment_tables_of_the_reference(tmp_path):
    # update_nhdp_from_alignment (impl/nanopore_hdp.c:206-297) on the reference's two table shapes: the 15-column alignment of its HDP
    # tests (tests/test_alignments/simple_alignment.tsv) and the 4-column assignments file signalMachine -s 2 writes
    aln = str(tmp_path / "simple_alignment.tsv")
    open(aln, "w").write(gzip.open(os.path.join(HDP, "simple_alignment.tsv.gz"), "rt").read())
    flat = sa.HdpState.new(sa.HDP_LAYOUT_FLAT, "ACGT", 6, (0.0, 100.0, 100), NIG, gamma=[4.0, 20.0])
    n = flat.pass_assignment_file(aln)
    rows = [ln.split() for ln in open(aln).read().split("\n") if ln.strip()]
    assert n == len(rows) == 1907 == flat.info.n_data
    assert np.array_equal(flat.array("data"), np.array([r[13] for r in rows], dtype=np.float64))
    assert np.array_equal(flat.array("data_dp"), np.array([flat.kmer_dp(r[9]) for r in rows]))
    assert flat.info.n_observed == len({r[9] for r in rows}) + 1
    asg = str(tmp_path / "assignments.tsv")
    open(asg, "w").write(gzip.open(os.path.join(HDP, "d6160b0b-a35e-43b5-947f-adaa1abade28.sm.assignments.tsv.gz"), "rt").read())
    nt = flat.pass_assignment_file(asg, strand="t")
    assert nt == 17350 == flat.info.n_data
    with pytest.raises(sa.SaError):
        flat.pass_assignment_file(asg, strand="c")                 # no complement rows in a 1-D file
    with pytest.raises(sa.SaError) as ei:
        sa.HdpState.new(sa.HDP_LAYOUT_FLAT, "ACG", 6, (0.0, 100.0, 100), NIG, gamma=[4.0, 20.0]).pass_assignment_file(asg)
    assert ei.value.code == -4                                      # a T in a k-mer: outside this HDP's alphabet
    # hdpHmm_loadFromFile's hand-over: the assignments of an expectations file
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    m = sa.Model.create(alpha, k, t10, tab)
    h = sa.Hmm.create(m, sa.HMM_HDP, 0.01, 0.0)
    for r in rows[:50]:
        h.add_assignment(r[9], float(r[13]))
    p = str(tmp_path / "x.expectations.tsv")
    h.write(p)
    kmers, events = sa.Hmm.load(p, sa.HMM_HDP).assignments()
    flat.pass_assignments(kmers, events)
    assert flat.info.n_data == 50 and np.allclose(flat.array("data"), [float(r[13]) for r in rows[:50]], atol=5e-7)


def test_a_plain_tree_of_more_than_67_processes_round_trips(tmp_path):
    # sa_hdp_state_new_tree writes its state under a placeholder header (alphabet "A", k = 1): the loader's bound on the number of
    # processes (3 A^k + 64 for a NanoporeHDP) must not apply to it
    parents = [-1] + [0] * 9 + [1 + (i % 9) for i in range(190)]
    s = sa.HdpState.new_tree(parents, 3, (-10.0, 10.0, 50), NIG, gamma=[1, 1, 1])
    assert s.info.num_dps == 200
    a, b = str(tmp_path / "tree.hdp"), str(tmp_path / "tree2.hdp")
    s.write(a)
    t = sa.HdpState(a)
    assert t.info.num_dps == 200 and t.array("dp_parent").tolist() == parents
    t.write(b)
    assert open(a).read() == open(b).read()
