"""Pins the CPU oracle against the reference's OWN known-answer tests.

Each test names the upstream test it restates (paths relative to the signalAlign tree).  The literal
inputs and expected values below are the data those tests hold.
"""
import math
import os

import numpy as np
import pytest


def test_logadd_matches_reference_bound(oracle):
    # tests/signalPairwiseAlignerTest.c:115-125  test_logAdd: |exp(logAdd(log i, log j)) - (i+j)| < 1e-3
    rng = np.random.default_rng(5)
    L = oracle.lib()
    for _ in range(100000):
        i, j = rng.random(), rng.random()
        if i == 0 or j == 0:
            continue
        l = math.exp(L.sao_log_add(math.log(i), math.log(j)))
        assert i + j - 0.001 < l < i + j + 0.001


def test_logadd_edge_semantics(oracle):
    # impl/pairwiseAligner.c:314-318: -inf is absorbing-neutral, cut-off at 7.5, symmetric
    L = oracle.lib()
    ninf = float("-inf")
    assert L.sao_log_add(ninf, -3.0) == -3.0
    assert L.sao_log_add(-3.0, ninf) == -3.0
    assert L.sao_log_add(ninf, ninf) == ninf
    assert L.sao_log_add(0.0, -7.5) == 0.0
    assert L.sao_log_add(0.0, -7.4999) > 0.0
    assert L.sao_log_add(-2.0, -1.0) == L.sao_log_add(-1.0, -2.0)
    # float-typed literals: first piece at x = 0 gives the float constant promoted to double
    assert L.sao_log_add(1.0, 1.0) == float(np.float32(0.693203116424741)) + 1.0


def test_kmer_id(oracle):
    # tests/nanoporeHdpTests.c:102-108  test_kmer_id
    L = oracle.lib()
    assert L.sao_kmer_id(b"AAAC", b"ACGT", 4, 4) == 1
    assert L.sao_kmer_id(b"AAAT", b"ACGT", 4, 4) == 3
    assert L.sao_kmer_id(b"AAAT", b"ACT", 3, 4) == 2
    assert L.sao_kmer_id(b"GGGG", b"ABCDEFG", 7, 4) == 7 ** 4 - 1
    assert L.sao_kmer_id(b"AACAA", b"ACGT", 4, 5) == 16
    assert L.sao_kmer_id(b"AANAA", b"ACGT", 4, 5) == -1  # the reference exits here


def test_bands(oracle):
    # tests/signalPairwiseAlignerTest.c:434-497  test_bands: anchors (1,0),(2,1),(3,3), lX=6, lY=5, expansion 2
    L, R = oracle.band([1, 2, 3], [0, 1, 3], 6, 5, 2)
    expect = [(0, 0), (-1, 1), (-2, 2), (-1, 3), (-2, 4), (-1, 3), (-2, 4), (-3, 3), (-2, 2), (-1, 3), (0, 2), (1, 1)]
    assert list(zip(L.tolist(), R.tolist())) == expect


def test_diagonal_parity_exception(oracle):
    # tests/signalPairwiseAlignerTest.c:499-541 test_diagonal
    L = oracle.lib()
    xL, yL, xU, yU = 10, 20, 30, 0
    assert L.sao_diagonal_check(xL + yL, xL - yL, xU - yU) == (xU - yU - (xL - yL)) // 2 + 1
    assert L.sao_diagonal_check(10, 5, 5) == -1   # parity
    assert L.sao_diagonal_check(10, 6, 4) == -1   # xmyR < xmyL


def test_get_split_points(oracle):
    # tests/signalPairwiseAlignerTest.c:363-432  test_getSplitPoints
    ms = 2000 * 2000
    sp = oracle.split_points([], [], 3000, 1000, ms, 0, 0)
    assert sp.tolist() == [[0, 0, 3000, 1000]]
    lX, lY = 20000, 25000
    assert oracle.split_points([], [], lX, lY, ms, 1, 1).tolist() == []
    assert oracle.split_points([], [], lX, lY, ms, 1, 0).tolist() == [[18000, 23000, lX, lY]]
    assert oracle.split_points([], [], lX, lY, ms, 0, 1).tolist() == [[0, 0, 2000, 2000]]
    assert oracle.split_points([], [], lX, lY, ms, 0, 0).tolist() == [[0, 0, 2000, 2000], [18000, 23000, lX, lY]]
    ax = [2000, 4002, 5000, 8000, 9000, 10000, 15000, 16000]
    ay = [2000, 4001, 5000, 6000, 9000, 14000, 15000, 16000]
    sp = oracle.split_points(ax, ay, lX, lY, ms, 0, 0)
    assert sp.tolist() == [[0, 0, 3001, 3001], [3002, 3001, 9500, 11001], [9501, 12000, 12001, 14500],
                           [13000, 14501, 18000, 18001], [18001, 23000, 20000, 25000]]


def test_hdcell_path_expansion(oracle, golden):
    # tests/signalPairwiseAlignerTest.c:543-568 test_hdCellConstruct / WorstCase: X -> C/E/O, 6-mers, ACEGOT
    m = oracle.Model.from_file(os.path.join(golden, "models", "testModelR73_acegot_template.model"))
    amb = oracle.ambig_map({"X": "CEO"})
    n, ids = m.expand_paths("ATGXAX", amb)
    assert n == 9
    assert ids[0] == m.kmer_id("ATGCAC") and ids[8] == m.kmer_id("ATGOAO")
    n, ids = m.expand_paths("XXXXXX", amb)
    assert n == 729
    assert ids[0] == m.kmer_id("CCCCCC") and ids[728] == m.kmer_id("OOOOOO")
    # order: left-to-right over positions, inner loop over replacement letters (impl/pairwiseAligner.c:749-778)
    n, ids = m.expand_paths("ATGXAX", amb)
    assert ids[1] == m.kmer_id("ATGCAE") and ids[3] == m.kmer_id("ATGEAC")


def test_default_ambig_table(oracle, golden):
    # tests/signalPairwiseAlignerTest.c:781-787 test_create_ambig_bases: X -> ACGT (impl/pairwiseAligner.c:32-65)
    m = oracle.Model.from_file(os.path.join(golden, "models", "testModelR9_5mer_acgt_template.model"))
    n, ids = m.expand_paths("AXGTA")
    assert n == 4 and ids == [m.kmer_id("A%sGTA" % c) for c in "ACGT"]
    n, ids = m.expand_paths("ACGTA")
    assert n == 1


def test_load_pore_model(oracle, golden):
    # tests/stateMachineTests.c:233-291 test_poreModel: table i == i, gapY sd == 1.75 * match sd
    k, alpha = 5, "ACGT"
    n = 5 * len(alpha) ** k
    t10 = np.arange(10, dtype=np.float64) + 1.0
    m = oracle.Model(alpha, k, t10, np.arange(n, dtype=np.float64))
    tab = m.match_table()
    assert np.array_equal(tab, np.arange(n, dtype=np.float64))


SY6 = [58.743435, 0.887833, 0.0571, 0.0,
       53.604965, 0.816836, 0.0571, 0.1,
       58.432015, 0.735143, 0.0571, 0.2,
       63.684352, 0.795437, 0.0571, 0.3,
       58.921430, 0.812959, 0.0571, 0.4,
       59.895882, 0.740952, 0.0571, 0.5,
       61.684303, 0.722332, 0.0571, 0.67]

SY5 = [70.0423375640843, 2.1070814631739, 0.0571, 0.0,
       73.7087073662952, 1.90162684687837, 0.0571, 0.1,
       105.375581864011, 2.87252862011704, 0.0571, 0.2,
       82.9620934477158, 2.38320603353748, 0.0571, 0.3,
       84.6977645711335, 3.08486975249442, 0.0571, 0.4,
       58.0551144225027, 2.52297561817531, 0.0571, 0.5,
       94.337668063878, 1.9731952395105, 0.0571, 0.67]


def test_sm3_diagonal_dp_calculations(oracle, golden):
    # tests/stateMachineTests.c:441-565  test_sm3_diagonalDPCalculations
    # ACGATALGGACAT (L -> C/E/O), 7 literal events, R7.3 ACEGOT 6-mer model, getStateMachine3 emissions
    m = oracle.Model.from_file(os.path.join(golden, "models", "testModelR73_acegot_template.model"),
                               emission=oracle.EM_TWODIST)
    ev = np.array(SY6, dtype=np.float64).reshape(7, 4)
    tF, tB, diag, pairs = oracle.kat_unbanded(m, "ACGATALGGACAT", ev, 0.2)
    assert abs(tF - tB) < 0.001
    assert np.all(np.abs(diag - tF) < 0.01)
    allowed = {(0, 0), (1, 1), (2, 2), (3, 3), (4, 3), (5, 4), (6, 5), (7, 6)}
    assert len(pairs) == 14
    assert {(int(p["x"]), int(p["y"])) for p in pairs} <= allowed
    assert np.all(pairs["prob_e7"] > 0) and np.all(pairs["prob_e7"] <= 10000000)


def test_sm3_5mer_diagonal_dp_calculations(oracle, golden):
    # tests/stateMachineTests.c:567-698  test_sm3_5merDiagonalDPCalculations
    m = oracle.Model.from_file(os.path.join(golden, "models", "testModelR9_5mer_acgt_template.model"),
                               emission=oracle.EM_TWODIST)
    ev = np.array(SY5, dtype=np.float64).reshape(7, 4)
    tF, tB, diag, pairs = oracle.kat_unbanded(m, "ACGATATGGACAT", ev, 0.2)
    assert abs(tF - tB) < 0.001
    assert np.all(np.abs(diag - tF) < 0.01)
    allowed = {(0, 0), (1, 1), (2, 2), (3, 3), (5, 4), (6, 5), (8, 6)}
    assert len(pairs) == 7
    assert {(int(p["x"]), int(p["y"])) for p in pairs} <= allowed


def test_emission_pdfs(oracle):
    # tests/signalPairwiseAlignerTest.c:75-105 test_stateMachine3EmissionsPdfs, through a 1-kmer-wide view:
    # a MeanOnly model with scale=1, shift=0, var=1 reduces to the plain log-Gaussian pdf.
    k, alpha = 1, "ACGT"
    table = np.zeros(20)
    mu, sd = 3.0, 0.7
    table[0::5] = mu
    table[1::5] = sd
    table[2::5] = 1.0
    table[3::5] = 1.0
    table[4::5] = 1.0
    t10 = np.array([0.79, 0.19, 0.013, 0.8, 0.19, 0.0, 0.98, 1e-9, 0.013, 0.0])
    m = oracle.Model(alpha, k, t10, table)
    # one k-mer, one event, no anchors: posterior of the single match must be high when the event fits
    pairs = oracle.align(m, "A", [3.05], [], [], oracle.default_params(threshold=0.0))
    assert len(pairs) == 1 and pairs[0]["x"] == 0 and pairs[0]["y"] == 0


def test_nanopore_read_fixtures(oracle, golden):
    # tests/stateMachineTests.c:224-231 test_checkTestNanoporeReads
    r = oracle.parse_npread(os.path.join(golden, "npReads", "ZymoC_ch_1_file1.npRead"))
    assert (r["read_length"], r["n_template_events"], r["n_complement_events"]) == (950, 799, 670)
    assert (r["template_read_length"], r["complement_read_length"]) == (879, 766)
    # tests/signalPairwiseAlignerTest.c:209-216 test_1dNanoporeRead
    r = oracle.parse_npread(os.path.join(golden, "npReads", "r9p4_oneD.npRead"))
    assert r["twoD"] == 0 and r["n_template_events"] == 10922


def test_oracle_reproduces_committed_expected_outputs(oracle, golden):
    GOLDEN = golden
    # tests/golden/expected/*.npz were written by tests/golden/make_expected.py; the oracle must keep producing them
    # bit for bit (guards the checker itself against drift; the GPU suite compares the HIP path with the same files)
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_expected", os.path.join(GOLDEN, "make_expected.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    for name, npread, model_path, nhdp in mk.CASES:
        want = np.load(os.path.join(GOLDEN, "expected", name + ".npz"))
        got = mk.compute(name, npread, model_path, nhdp)
        for key in ("scale", "shift", "var"):
            assert float(got[key]) == float(want[key]), (name, key)
        for key in ("ax", "ay", "x", "y", "path", "kmer_id", "prob_e7"):
            assert np.array_equal(got[key], want[key]), (name, key)
    for name in mk.ZYMO_CASES:
        want = np.load(os.path.join(GOLDEN, "expected", name + ".npz"))
        got = mk.compute_zymo(name)
        for key in ("x", "y", "path", "kmer_id", "prob_e7"):
            assert np.array_equal(got[key], want[key]), (name, key)
    # the HDP case is the reference's own job and count (tests/stateMachineTests.c:912)
    assert len(np.load(os.path.join(GOLDEN, "expected", "zymo_lastz_hdp.npz"))["x"]) == 1217


def test_scale_params_from_strand_read_and_drift(oracle, golden):
    # tests/stateMachineTests.c:408-421 test_nanoporeScaleParamsFromStrandRead: the parameters estimated from the
    # strand read's own event map land within 5 % (scale, shift) and 50 % (var) of the ones in the .npRead header;
    # :423-440 test_adjustForDrift: afterwards every event mean is the original minus drift * start time, exactly.
    def pct(a, b):
        return 100.0 * abs(a - b) / abs(b)
    r = oracle.parse_npread(os.path.join(golden, "npReads", "ZymoC_ch_1_file1.npRead"))
    om = oracle.Model.from_file(os.path.join(golden, "models", "testModelR73_acegot_template.model"))
    ev = r["template_events"].copy()
    est = oracle.estimate_params(om, r["template_strand_event_map"], ev, r["template_read"])
    hdr = r["template_params"]
    assert pct(est["scale"], hdr["scale"]) < 5.0
    assert pct(est["shift"], hdr["shift"]) < 5.0
    assert pct(est["var"], hdr["var"]) < 50.0
    r9 = oracle.parse_npread(os.path.join(golden, "npReads", "c2925_ecoli_ch34_read1023.npRead"))
    om9 = oracle.Model.from_file(os.path.join(golden, "models", "testModelR9_5mer_acgt_template.model"))
    ev9 = r9["template_events"].copy()
    est9 = oracle.estimate_params(om9, r9["template_strand_event_map"], ev9, r9["template_read"])
    orig = r9["template_events"]
    assert np.array_equal(ev9[:, 0], orig[:, 0] - est9["drift"] * orig[:, 3])
    assert np.array_equal(ev9[:, 1:], orig[:, 1:])


def test_path_legal_transitions_and_substituted_kmers(oracle, golden):
    # tests/variableOrderPairwiseAlignerTests.c:107-123 test_pathLegalTransitions: AAETTT -> AETTTC is legal,
    # AAETTT -> ACTTTC is not (k-1 suffix must equal k-1 prefix; the NULL k-mer is legal with everything);
    # :170-180 test_substitutedKmers: ATGXAX with the pattern "CE" is ATGCAE.
    m = oracle.Model.from_file(os.path.join(golden, "models", "testModelR73_acegot_template.model"))
    A, k = m.n_alpha, m.k

    def legal(a, b):  # the arithmetic the kernels use (sa_hip.hip:legal_step)
        if a is None or b is None:
            return True
        return m.kmer_id(a) % A ** (k - 1) == m.kmer_id(b) // A
    assert legal(None, "AAETTT") and legal("AAETTT", None)
    assert legal("AAETTT", "AETTTC")
    assert not legal("AAETTT", "ACTTTC")
    n, ids = m.expand_paths("ATGXAX", oracle.ambig_map({"X": "CE"}))
    assert n == 4 and ids[1] == m.kmer_id("ATGCAE")
    # and the DP itself obeys it: an ambiguous reference only yields pairs whose path k-mers chain legally
    r = oracle.parse_npread(os.path.join(golden, "npReads", "ZymoC_ch_1_file1.npRead"))
    ref = r["template_read"][:120].replace("C", "X", 3)
    ev = r["template_events"][:int(r["template_strand_event_map"][119])]
    pairs = oracle.align(m, ref, ev, [], [], oracle.default_params(threshold=0.005), ambig=oracle.ambig_map({"X": "CE"}))
    by_x = {}
    for p in pairs:
        by_x.setdefault(int(p["x"]), set()).add(int(p["kmer_id"]))
    checked = 0
    for x, ids_x in by_x.items():
        if x + 1 in by_x:
            # every reported k-mer at x+1 has at least one legal predecessor among the k-mers of position x
            pre = set(i % A ** (k - 1) for i in m.expand_paths(ref[x:x + k], oracle.ambig_map({"X": "CE"}))[1])
            for j in by_x[x + 1]:
                assert j // A in pre
                checked += 1
    assert checked >= 3


# ---------------------------------------------------------------------------------------------------------------------
# Whole-read known answers of the reference: ZymoC_ch_1_file1.npRead (799 template events) x ZymoRef.txt
# (tests/stateMachineTests.c:842-983).  Anchors: the committed output of the reference's lastz subprocess, converted by
# tests/zymo_wholeread.py.  These are the only reference-held vectors over a real read; they pin the banded driver
# (A12/A13), the split/anchor handling (A15/A16), the ambiguity expansion (A5) and the HDP emission (A10) of the oracle.
# ---------------------------------------------------------------------------------------------------------------------
def _zymo(oracle, golden):
    import zymo_wholeread as z
    r = z.read_fixture()
    d = oracle.parse_model_file(os.path.join(golden, "models", "testModelR73_acegot_template.model"))
    ax, ay = z.remapped_anchors()
    b = z.BANDING
    p = oracle.Params(b["threshold"], b["expansion"], b["trace_back"], b["min_diags"], b["split"], z.TRIM)
    return z, r, d, ax, ay, p


def test_zymo_whole_read_without_banding_1076(oracle, golden):
    # tests/stateMachineTests.c:855-868: getAlignedPairsWithoutBanding, scaled R7.3 model (loadScaledStateMachine3 :69-79,
    # emissions_signal_scaleModel impl/stateMachine.c:743-779), threshold 0.01 -> exactly 1076 pairs
    z, r, d, ax, ay, p = _zymo(oracle, golden)
    m = oracle.Model(d["alphabet"], d["k"], d["transitions10"], z.scaled_table(d["table5"], r["template_params"]),
                     emission=oracle.EM_TWODIST)
    tF, tB, diag, pairs = oracle.kat_unbanded(m, r["ref"], r["template_events"], 0.01)
    assert len(pairs) == z.N_PAIRS_GAUSS
    assert abs(tF - tB) < 0.01
    lX = len(r["ref"]) - 5
    assert pairs["x"].min() >= 0 and pairs["x"].max() < lX and pairs["y"].min() >= 0 and pairs["y"].max() < 799


def test_zymo_whole_read_banded_1076_scaled_and_descaled(oracle, golden):
    # tests/stateMachineTests.c:851-852: getAlignedPairsUsingAnchors (ragged 1, 1) with the scaled model and with the
    # descaled one (getStateMachine3_descaled + scaleNoise, impl/stateMachine.c:1739-1755): 1076 == the un-banded count
    z, r, d, ax, ay, p = _zymo(oracle, golden)
    tp = r["template_params"]
    assert len(ax) == 39
    m = oracle.Model(d["alphabet"], d["k"], d["transitions10"], z.scaled_table(d["table5"], tp), emission=oracle.EM_TWODIST)
    assert len(oracle.align(m, r["ref"], r["template_events"], ax, ay, p, ragged=(1, 1))) == z.N_PAIRS_GAUSS
    m2 = oracle.Model(d["alphabet"], d["k"], d["transitions10"], d["table5"], emission=oracle.EM_TWODIST_DESCALED)
    m2.set_read_params(tp["scale"], tp["shift"], tp["var"])
    m2.scale_noise(tp["scale_sd"], tp["var_sd"])
    assert len(oracle.align(m2, r["ref"], r["template_events"], ax, ay, p, ragged=(1, 1))) == z.N_PAIRS_GAUSS


def test_zymo_whole_read_degenerate_nucleotides(oracle, golden):
    # tests/stateMachineTests.c:920-983 test_DegenerateNucleotides: every C of the reference replaced by C / E / O keeps
    # 1076 pairs; replaced by the ambiguity letter L (C, E or O: three paths per C in the k-mer) gives 7349.  Ragged 0, 0.
    z, r, d, ax, ay, p = _zymo(oracle, golden)
    tp = r["template_params"]
    m = oracle.Model(d["alphabet"], d["k"], d["transitions10"], d["table5"], emission=oracle.EM_TWODIST_DESCALED)
    m.set_read_params(tp["scale"], tp["shift"], tp["var"])
    m.scale_noise(tp["scale_sd"], tp["var_sd"])
    for letter, want in z.N_PAIRS_DEGENERATE.items():
        pairs = oracle.align(m, r["ref"].replace("C", letter), r["template_events"], ax, ay, p, ragged=(0, 0))
        assert len(pairs) == want, (letter, len(pairs))
        # checkAlignedPairs[WithOverlap] (:154-221): coordinates inside the matrix, probabilities in (0, PROB_1]
        assert pairs["prob_e7"].min() > 0 and pairs["prob_e7"].max() <= 10000000
        if letter != "L":
            assert len({(int(a), int(b)) for a, b in zip(pairs["x"], pairs["y"])}) == want


def test_zymo_whole_read_hdp_1217(oracle, golden):
    # tests/stateMachineTests.c:902-918 test_sm3Hdp_getAlignedPairsWithBanding: the bundled single-level HDP, threshold 0.1,
    # events "descaled" by nanopore_descaleNanoporeRead (see zymo_wholeread.hdp_test_events) -> exactly 1217 pairs
    z, r, d, ax, ay, p = _zymo(oracle, golden)
    tp = r["template_params"]
    m = oracle.Model(d["alphabet"], d["k"], d["transitions10"], d["table5"], emission=oracle.EM_HDP)
    m.load_hdp(os.path.join(golden, "models", "templateSingleLevelFixed.nhdp"))
    m.set_read_params(tp["scale"], tp["shift"], tp["var"])
    p.threshold = 0.1
    pairs = oracle.align(m, r["ref"], z.hdp_test_events(r), ax, ay, p, ragged=(1, 1))
    assert len(pairs) == z.N_PAIRS_HDP_AT_0p1
