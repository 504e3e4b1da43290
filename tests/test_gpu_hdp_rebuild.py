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


# ---------------------------------------------------------------------------------------------------------------------------------
# The Gibbs sweeps (sa_hdp_state_gibbs: host code around the kernels above) -- the reference's own tests of this code are
# properties, mirrored here with its data (tests/golden/hdp/: tests/test_hdp/data.txt, dps.txt, tests/test_alignments/
# simple_alignment.tsv, tests/test_assignment_files/d6160b0b-...).  PARITY UNPINNED: the draws come from this library's own
# seeded generator.
# ---------------------------------------------------------------------------------------------------------------------------------
import gzip
import os
import subprocess

from signalalign_amd import synth

HDP_DATA = os.path.join(cases.GOLDEN, "hdp")


def _integral(grid, y):
    return float(np.sum(0.5 * (y[..., 1:] + y[..., :-1]) * np.diff(grid), axis=-1).max()), \
        float(np.sum(0.5 * (y[..., 1:] + y[..., :-1]) * np.diff(grid), axis=-1).min())


def _write_load_write(s, tmp_path, tag):
    a, b = str(tmp_path / (tag + "_a.hdp")), str(tmp_path / (tag + "_b.hdp"))
    s.write(a)
    c = sa.HdpState(a)
    c.write(b)
    assert open(a).read() == open(b).read(), tag
    return c


def test_serialization_through_a_sampling_run_of_the_reference_test_hdp(tmp_path):
    """tests/nanoporeHdpTests.c:272-460 test_serialization: the 8-process, depth-3 HDP with a Gamma prior on its concentration
    parameters, the reference's 39 877 data points; serialise -> deserialise -> serialise gives the same file with data, after
    execute_gibbs_sampling(10, 10, 10) and after finalize_distributions -- and add_hdp_copy_tests' comparisons hold trivially."""
    data = np.array(gzip.open(os.path.join(HDP_DATA, "test_hdp_data.txt.gz"), "rt").read().split(), dtype=np.float64)
    dps = np.array(gzip.open(os.path.join(HDP_DATA, "test_hdp_dps.txt.gz"), "rt").read().split(), dtype=np.int64)
    keep = dps != 4
    s = sa.HdpState.new_tree([-1, 0, 0, 1, 1, 1, 2, 2], 3, (-10.0, 10.0, 250), (0.0, 1.0, 2.0, 10.0),
                             gamma_alpha=[1.0, 1.0, 2.0], gamma_beta=[0.2, 0.2, 0.1])
    s.pass_data(data[keep], dps[keep])
    s.gibbs(10, 10, 10, seed=7)
    assert s.samples_taken() == 10 and s.info.splines_finalized == 0
    c = _write_load_write(s, tmp_path, "sampled")
    # the factor tree moved: more than the one base factor init_factors made, every data point still under a factor of its own DP
    # (sa_hdp_state_load checks exactly that, and the children counts, on the way in)
    ft = c.array("f_type")
    assert (ft == 0).sum() >= 1 and (ft == 2).sum() == keep.sum() and c.info.n_factors > 1 + 6 + keep.sum() - 1
    assert c.array("dp_num_factor_children")[4] == 0 and not c.array("observed")[4]
    # ten samples, ten iterations apart, come long before the first sweep over 39 877 data points ends (as in the reference's test):
    # the concentration parameters still stand at their prior's mean.  A run across two sweeps samples them.
    assert np.array_equal(s.array("gamma"), [5.0, 5.0, 20.0])
    s.gibbs(2, 60000, 15000, seed=8)
    g = s.array("gamma")
    assert s.samples_taken() == 12 and np.all(g > 0) and np.all(g != [5.0, 5.0, 20.0])
    c = _write_load_write(s, tmp_path, "sampled_twice")
    assert np.array_equal(c.array("gamma"), g)
    s.finalize()
    assert s.info.splines_finalized == 1
    c = _write_load_write(s, tmp_path, "finalized")
    grid, post = s.array("grid"), s.array("post")
    assert np.array_equal(c.array("post"), post) and np.array_equal(c.array("slope"), s.array("slope"))
    # densities: non-negative, and they integrate to one over a grid that holds the data (|x| < 10 for all but a handful)
    # (the data's sd is 3; the inner DPs have few factors under them, so the wide base distribution -- a t of two degrees of
    # freedom and scale 4.5 -- keeps weight gamma / (gamma + children) there and part of its tails lies outside the grid)
    rows = s.array("row_of_dp")
    hi, lo = _integral(grid, post)
    assert post.min() >= 0.0 and 0.9 < lo and hi < 1.01
    hi, lo = _integral(grid, post[rows[[3, 5, 6, 7]]])
    assert 0.97 < lo and hi < 1.01
    # the leaves' densities follow their own data: the grid mean of each observed leaf lies near its sample mean
    # (two sweeps away from init_factors' single cluster: near, not at -- the means order as the data's do)
    means = {}
    for d in (3, 5, 6, 7):
        y = post[rows[d]]
        means[d] = float(np.sum(grid * y) / np.sum(y))
        assert abs(means[d] - data[keep][dps[keep] == d].mean()) < 0.6, (d, means[d])
    assert means[5] > means[3] > means[7] > means[6]
    with pytest.raises(sa.SaError):
        s.gibbs(1, 0, 1)                                                     # finalised: new data first
    with pytest.raises(sa.SaError):
        s.finalize()


def test_gibbs_sweeps_separate_two_leaves():
    """What the sampler is for, on data with a known answer: two leaves under one base DP, 400 points each at -5 +- 0.5 and +5 +- 0.5.
    After twenty sweeps each leaf's density sits on its own data, the base DP's is the even mixture of the two, and a handful of base
    factors carry everything."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.normal(-5.0, 0.5, 400), rng.normal(5.0, 0.5, 400)])
    d = np.concatenate([np.full(400, 1), np.full(400, 2)])
    s = sa.HdpState.new_tree([-1, 0, 0], 2, (-10.0, 10.0, 401), (0.0, 0.05, 2.0, 1.0), gamma=[1.0, 1.0])
    s.pass_data(x, d)
    s.gibbs(50, 20 * 900, 50, seed=1)
    assert (s.array("f_type") == 0).sum() <= 8
    s.finalize()
    grid, post, rows = s.array("grid"), s.array("post"), s.array("row_of_dp")
    mean_of = lambda y: float(np.sum(grid * y) / np.sum(y))
    sd_of = lambda y: float(np.sqrt(np.sum((grid - mean_of(y)) ** 2 * y) / np.sum(y)))
    assert abs(mean_of(post[rows[1]]) + 5.0) < 0.15 and abs(mean_of(post[rows[2]]) - 5.0) < 0.15
    assert sd_of(post[rows[1]]) < 1.0 and sd_of(post[rows[2]]) < 1.0
    base = post[rows[0]]
    assert abs(mean_of(base)) < 1.0 and sd_of(base) > 4.0
    left = float(np.sum(0.5 * (base[1:] + base[:-1]) * np.diff(grid) * (grid[1:] <= 0)))
    assert 0.3 < left < 0.7
    # the density at a leaf's own data against the other leaf's: orders of magnitude
    i_lo, i_hi = int(np.argmin(abs(grid + 5.0))), int(np.argmin(abs(grid - 5.0)))
    assert post[rows[1], i_lo] > 50 * post[rows[1], i_hi] and post[rows[2], i_hi] > 50 * post[rows[2], i_lo]


def test_nhdp_from_the_reference_alignment_table(tmp_path):
    """tests/hdpTests.c:233-255 test_nhdp_distrs / tests/nanoporeHdpTests.c:462-480 test_nhdp_serialization: flat_hdp_model("ACGT", 4,
    6, 4.0, 20.0, 0.0, 100.0, 100, testModelR73_acegot_template.model), update_nhdp_from_alignment(simple_alignment.tsv),
    execute_nhdp_gibbs_sampling(100, 0, 1), finalize; serialise / deserialise; the same seed gives the same file."""
    aln = str(tmp_path / "simple_alignment.tsv")
    open(aln, "w").write(gzip.open(os.path.join(HDP_DATA, "simple_alignment.tsv.gz"), "rt").read())
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_R73)
    nig = sa.hdp_nig_params_from_table(tab)                                  # normal_inverse_gamma_params_from_minION

    def build(seed):
        s = sa.HdpState.new(sa.HDP_LAYOUT_FLAT, "ACGT", 6, (0.0, 100.0, 100), nig, gamma=[4.0, 20.0])
        assert s.pass_assignment_file(aln) == 1907
        s.gibbs(100, 0, 1, seed=seed)
        s.finalize()
        return s
    s = build(3)
    c = _write_load_write(s, tmp_path, "nhdp")
    assert np.array_equal(c.array("post"), s.array("post"))
    p2 = str(tmp_path / "again.nhdp")
    build(3).write(p2)
    assert open(p2).read() == open(str(tmp_path / "nhdp_a.hdp")).read()      # deterministic given the seed
    p3 = str(tmp_path / "other.nhdp")
    build(4).write(p3)
    assert open(p3).read() != open(p2).read()
    grid, post, rows = s.array("grid"), s.array("post"), s.array("row_of_dp")
    hi, lo = _integral(grid, post)
    assert post.min() >= 0.0 and lo > 0.9 and hi < 1.02
    # more data under a k-mer pulls its density towards them: a k-mer seen several times peaks nearer its own mean than the base DP does
    lines = [ln.split() for ln in open(aln).read().split("\n") if ln.strip()]
    by_kmer = {}
    for t in lines:
        by_kmer.setdefault(t[9], []).append(float(t[13]))
    kmer, vals = max(by_kmer.items(), key=lambda kv: len(kv[1]))
    assert len(vals) >= 3
    y_leaf, y_base = post[rows[s.kmer_dp(kmer)]], post[rows[s.info.base_dp]]
    mean_of = lambda y: float(np.sum(grid * y) / np.sum(y))
    assert abs(mean_of(y_leaf) - np.mean(vals)) < abs(mean_of(y_base) - np.mean(vals))


def test_rebuilt_hdp_aligns(tmp_path):
    """The whole loop of trainModels.py --hdp on the reference's fixtures, through the two executables: (1) buildHdpUtil builds a
    flat ACGT 6-mer HDP from the bundled assignments file (tests/test_assignment_files/d6160b0b-...: 17 350 template rows of an R9.4
    read) and the library's aligner loads it and aligns reads with it; (2) updateHdpFromAssignments: the bundled HDP, the Zymo read's
    own assignments (sa_expect_batch -> the expectations file signalMachine writes -> sa_hmm_load), Gibbs, finalise, serialise -- and the
    rebuilt file aligns the Zymo read again."""
    import zymo_wholeread as z
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "signalalign_amd", "bin", "buildHdpUtil")
    asg = str(tmp_path / "assignments.tsv")
    open(asg, "w").write(gzip.open(os.path.join(HDP_DATA, "d6160b0b-a35e-43b5-947f-adaa1abade28.sm.assignments.tsv.gz"), "rt").read())
    out = str(tmp_path / "template.singleLevelFixedCanonical.nhdp")
    cmd = [tool, "--verbose", "-p", "14", "-v", out, "-w", "None", "-l", asg, "-a", "6", "-n", "200", "-I", "20000", "-t", "100",
           "-s", "40", "-e", "140", "-k", "400", "--oneD", "-C", "None", "-T", cases.MODEL_6MER, "-B", "1", "-M", "1", "-L", "1", "-b", "ACGT"]
    pr = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-2000:]
    assert "Beginning sweep" in pr.stderr and "Serializing template to" in pr.stderr
    s = sa.HdpState(out)
    assert (s.info.splines_finalized, s.info.n_data, s.info.alphabet_size, s.info.kmer_length) == (1, 17350, 4, 6)
    hi, lo = _integral(s.array("grid"), s.array("post"))
    assert lo > 0.9 and hi < 1.02
    # the aligner takes the rebuilt file: reads of the R9.4 6-mer model, HDP emissions against Gaussian emissions -- the HDP saw real
    # R9.4 events of the same pore model (about four per k-mer: broad densities, more pairs above the threshold), so the two
    # alignments mostly agree on where an event belongs
    hd = sa.Model.load(cases.MODEL_6MER, out)
    hd.set_to_hdp_expected_values()
    ga = sa.Model.load(cases.MODEL_6MER)
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 3, 1200, 880)
    p = sa.default_params(threshold=0.2)
    bh, bg = sa.Batch(hd, p, jobs), sa.Batch(ga, p, jobs)
    bh.run(); bg.run()
    for j in range(3):
        h_ = {(int(q["x"]), int(q["y"])) for q in bh.pairs(j)}
        g_ = {(int(q["x"]), int(q["y"])) for q in bg.pairs(j)}
        assert len(h_) > 600 and len(h_ & g_) >= 0.5 * len(g_), (j, len(h_), len(g_), len(h_ & g_))
    # ... and the HIP path agrees with the CPU restatement on this DENSE model (3183 observed processes x 400 grid points: most k-mers
    # read a row of their own, none is shared by a quarter of them -- bench.py --workload hdp_dense is this model): 1e-5 on a posterior
    from oracle import sa_oracle_py as oracle
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    om = oracle.Model(alpha, k, t10, tab)
    om.load_hdp(out)
    om.set_to_hdp_expected_values()
    op = cases.oracle_params(oracle, p)
    worst = 0
    for j in range(2):
        exp = cases.oracle_pairs(oracle, om, jobs[j], op)
        w, lonely = cases.compare_pairs(bh.pairs(j), exp, 100, p.threshold)
        worst = max(worst, w)
        assert lonely <= 2 and cases.same_order(bh.pairs(j), exp), j
    assert worst <= 100
    bh.close(); bg.close()
    # (2) the Zymo read under the bundled HDP -> its assignments -> the HDP updated from them
    r = z.read_fixture()
    ax, ay = z.remapped_anchors()
    tp = r["template_params"]
    m = sa.Model.load(cases.MODEL_R73, cases.NHDP)
    b = z.BANDING
    pz = sa.default_params(threshold=0.1, expansion=b["expansion"], trace_back=b["trace_back"], min_diags=b["min_diags"], split=b["split"])
    job = dict(ref=r["ref"], events=z.hdp_test_events(r), ax=ax, ay=ay, scale=tp["scale"], shift=tp["shift"], var=tp["var"])
    trans, lik, assigns = sa.expect_batch(m, pz, [job])
    h = sa.Hmm.create(m, sa.HMM_HDP, pz.threshold, 0.0)
    h.add_expectations(trans[0], lik[0])
    for pos, ev in assigns[0]:
        h.add_assignment(job["ref"][int(pos):int(pos) + 6], job["events"][int(ev), 0])
    exp_file = str(tmp_path / "zymo.template.expectations.tsv")
    h.write(exp_file)
    n_as = h.view().n_assignments
    assert n_as > 300
    updated = str(tmp_path / "updated.nhdp")
    pr = subprocess.run([tool, "--updateFrom", cases.NHDP, "--expectations", exp_file, "-v", updated, "-n", "100", "-I", "3000", "-t", "30",
                         "--seed", "11"], capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-2000:]
    u = sa.HdpState(updated)
    assert u.info.n_data == n_as and u.info.splines_finalized == 1 and u.info.num_dps == 46657
    m2 = sa.Model.load(cases.MODEL_R73, updated)
    b2 = sa.Batch(m2, pz, [job])
    b2.run()
    got = b2.pairs(0)
    b2.close()
    assert 600 <= len(got) <= 2500                      # (1217 pairs under the bundled file; the rebuilt one saw this read only)
    assert got["x"].max() < len(r["ref"]) - 5 and got["y"].max() < 799


def test_rebuilt_densities_match_the_ones_the_reference_sampler_stored():
    """The one statistical pin the reference offers for its sampler: templateSingleLevelFixed.nhdp carries the data it was built from
    (750 events and their leaf DPs), its hyperparameters, and the densities the reference's own Gibbs run averaged.  The same data
    through this library's sweeps (flat ACEGOT 6-mer layout, the file's grid, base distribution and concentration parameters) give the
    same posterior predictive densities up to Monte-Carlo noise: L1 distance per observed DP (of a total mass of 1) median 0.0009,
    worst 0.0066 with 2000 samples, where two seeds of this sampler differ by 0.0004 / 0.0022 (probes/hdp_rebuild_vs_reference_file.py,
    profiles/r05_hdp_rebuild_vs_reference_file.txt).  A sampler that assigned, weighted or updated anything differently would not land
    inside 1 % of every one of 352 densities."""
    ref = sa.HdpState(cases.NHDP)
    i = ref.info
    grid, post_ref, rows_ref = ref.array("grid"), ref.array("post"), ref.array("row_of_dp")
    data, data_dp = ref.array("data"), ref.array("data_dp")
    obs = np.flatnonzero(ref.array("observed"))
    s = sa.HdpState.new(sa.HDP_LAYOUT_FLAT, "ACEGOT", 6, (float(grid[0]), float(grid[-1]), len(grid)), (i.mu, i.nu, i.alpha, i.beta),
                        gamma=[float(g) for g in ref.array("gamma")])
    s.pass_data(data, data_dp)
    s.gibbs(1200, 40 * len(data), 4 * len(data), seed=3)
    s.finalize()
    post, rows = s.array("post"), s.array("row_of_dp")
    assert np.array_equal(np.flatnonzero(s.array("observed")), obs) and len(obs) == 352
    d = np.abs(post[rows[obs]] - post_ref[rows_ref[obs]])
    l1 = np.sum(0.5 * (d[:, 1:] + d[:, :-1]) * np.diff(grid), axis=1)
    assert np.median(l1) < 0.003 and l1.max() < 0.015, (float(np.median(l1)), float(l1.max()))
    assert l1[list(obs).index(i.base_dp)] < 0.001


def test_a_transparent_middle_level_changes_nothing():
    """A check of the middle-level plumbing (middle factors are sampled through joint log-likelihoods, data-point factors through
    likelihoods: different code) that needs no reference: put a middle process between every leaf and the base process and give that
    level a concentration parameter of 1e8 -- every table a leaf opens then opens a table of its own in the middle process, which sits at
    the base process exactly as the leaf's table did without it.  The leaves' densities must agree with the two-level model's up to
    Monte-Carlo noise (two seeds of the two-level model differ by as much)."""
    data = np.array(gzip.open(os.path.join(HDP_DATA, "test_hdp_data.txt.gz"), "rt").read().split(), dtype=np.float64)[:1600]
    leaf = np.arange(1600) % 4
    grid, nig = (-12.0, 12.0, 121), (0.0, 0.2, 2.0, 4.0)

    def run(parents, depth, gamma, dp_of_leaf, seed):
        s = sa.HdpState.new_tree(parents, depth, grid, nig, gamma=gamma)
        s.pass_data(data, np.array([dp_of_leaf[q] for q in leaf], dtype=np.int64))
        s.gibbs(1500, 30 * len(data), 2 * len(data), seed=seed)
        s.finalize()
        post, rows = s.array("post"), s.array("row_of_dp")
        return np.array([post[rows[dp_of_leaf[q]]] for q in range(4)]), s.array("grid")
    two, g = run([-1, 0, 0, 0, 0], 2, [3.0, 2.0], {0: 1, 1: 2, 2: 3, 3: 4}, 1)
    two_b, _ = run([-1, 0, 0, 0, 0], 2, [3.0, 2.0], {0: 1, 1: 2, 2: 3, 3: 4}, 2)
    three, _ = run([-1, 0, 0, 0, 0, 1, 2, 3, 4], 3, [3.0, 1e8, 2.0], {0: 5, 1: 6, 2: 7, 3: 8}, 3)
    l1 = lambda a, b: np.sum(0.5 * (np.abs(a - b)[:, 1:] + np.abs(a - b)[:, :-1]) * np.diff(g), axis=1)
    noise, diff = l1(two, two_b), l1(two, three)
    print("transparent middle level: L1 two-level seed 1 vs seed 2 %s; two-level vs three-level %s" % (np.round(noise, 4), np.round(diff, 4)))
    assert noise.max() < 0.005 and diff.max() < 0.006, (noise, diff)   # (measured: 0.0008-0.0011 and 0.0014-0.0017)


def test_build_hdp_util_hard_coded_alphabets(tmp_path):
    """NanoporeHdpType 15-20: flat, fixed-gamma models whose alphabets the reference hard-codes (impl/nanopore_hdp.c:1160-1240,
    inc/stateMachine.h:25-30) -- trainModels.py composes `-p <type>` for them without -b; -b counts for unspecified types only."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "signalalign_amd", "bin", "buildHdpUtil")
    asg = str(tmp_path / "assignments.tsv")
    rows = gzip.open(os.path.join(HDP_DATA, "d6160b0b-a35e-43b5-947f-adaa1abade28.sm.assignments.tsv.gz"), "rt").read().splitlines()[:1500]
    open(asg, "w").write("\n".join(rows) + "\n")

    def run(ptype, extra=()):
        out = str(tmp_path / ("t%s.nhdp" % ptype))
        cmd = [tool, "-p", str(ptype), "-v", out, "-l", asg, "-a", "6", "-n", "10", "-I", "100", "-t", "5", "-s", "40", "-e", "140",
               "-k", "100", "--oneD", "-T", cases.MODEL_6MER, "-B", "1", "-L", "1"] + list(extra)
        pr = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        head = open(out).read(64).split("\n")[:3] if pr.returncode == 0 and os.path.exists(out) else None
        return pr, head
    for ptype, alphabet in ((15, "ACFGT"), (16, "ACGTbp"), (20, "ACGTabc")):
        pr, head = run(ptype)
        assert pr.returncode == 0, pr.stderr[-1000:]
        assert head == [str(len(alphabet)), alphabet, "6"], (ptype, head)
    pr, head = run(15, ("-b", "ACGT"))           # a known type keeps its own alphabet
    assert pr.returncode == 0 and head[1] == "ACFGT"
    pr, head = run(33)                           # an unspecified type has none
    assert pr.returncode != 0 and "alphabet" in pr.stderr
    pr, head = run(33, ("-b", "ACGT"))
    assert pr.returncode == 0 and head == ["4", "ACGT", "6"]
