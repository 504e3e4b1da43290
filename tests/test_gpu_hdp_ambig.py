"""HDP emissions together with ambiguous reference positions -- the reference's methylation-calling workflow
(`--sm3Hdp` with variant positions: stateMachine3HDP_cellCalculate impl/stateMachine.c:1371-1437 over the several paths of
hdCell_construct2 impl/pairwiseAligner.c:723-801, written out by writePosteriorProbsVC impl/signalMachine.c:161-232).

Every case goes through the C ABI (or the signalMachine command line) and is compared with the CPU restatement on the same
inputs.  Bar: 1e-5 absolute on a posterior (the HDP emission takes a logarithm: the device's and the C library's differ in
the last bit), rows on one side only within that tolerance of the threshold, same output order."""
import os
import subprocess

import numpy as np
import pytest

import signalalign_amd as sa
from signalalign_amd import synth

import sa_cases as cases

pytestmark = pytest.mark.gpu

TOL_E7 = 100
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "signalalign_amd", "bin", "signalMachine")


def _models(oracle):
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_R73)
    pm = sa.Model.load(cases.MODEL_R73, cases.NHDP)
    om = oracle.Model(alpha, k, t10, tab)
    om.load_hdp(cases.NHDP)
    pm.set_to_hdp_expected_values()
    om.set_to_hdp_expected_values()
    return pm, om


def _mark(ref, letter, every=1, what="CG"):
    """replaces the first base of every `every`-th occurrence of `what` by `letter` (not in the first / last ten bases)"""
    r, n = list(ref), 0
    for i in range(10, len(r) - 10):
        if ref[i:i + len(what)] == what:
            if n % every == 0:
                r[i] = letter
            n += 1
    return "".join(r)


def _jobs(pm, n_events, first, n=3):
    return cases.hdp_jobs(n, n_events, first, table5=pm.table5())


def _thin(job, step):
    q = dict(job)
    keep = np.zeros(len(q["ax"]), dtype=bool)
    keep[::step] = True
    q["ax"], q["ay"] = q["ax"][keep], q["ay"][keep]
    return q


@pytest.mark.parametrize("threshold", [0.01, 0.1])
def test_hdp_with_cpg_ambiguity_against_the_oracle(oracle, threshold):
    """every CpG cytosine is C or E (5-methylcytosine): 1-4 paths per cell; dense anchors, anchors thinned to a 41st (bands
    wider than a wave), a read of 40 events"""
    pm, om = _models(oracle)
    p = sa.default_params(threshold=threshold)
    op = cases.oracle_params(oracle, p)
    amb_p, amb_o = sa.default_ambig({"X": "CE"}), oracle.ambig_map({"X": "CE"})
    jobs = _jobs(pm, 1100, 31) + _jobs(pm, 40, 77, n=1)
    for j in jobs:
        j["ref"] = _mark(j["ref"], "X")
    assert sum(j["ref"].count("X") for j in jobs) > 40
    jobs.append(_thin(jobs[1], 41))
    b = sa.Batch(pm, p, jobs, ambig=amb_p)
    b.run()
    st = b.stats()
    got = [b.pairs(j) for j in range(len(jobs))]
    b.close()
    # several paths per cell: the ring kernels (they read one emission per cell-path from the plane k_emit_hdp_ring fills)
    # (a read without a CpG keeps one path per cell and the register kernels)
    assert st.n_ring_regions >= 4, (st.n_regions, st.n_fast_regions, st.n_ring_regions)
    worst, most_paths = 0, 0
    for j, job in enumerate(jobs):
        exp = cases.oracle_pairs(oracle, om, job, op, ambig=amb_o)
        assert len(exp) > (3.0 if threshold < 0.05 else 0.02) * len(job["events"])
        w, lonely = cases.compare_pairs(got[j], exp, TOL_E7, p.threshold)
        assert lonely <= max(4, len(exp) // 500)
        assert cases.same_order(got[j], exp)
        ek = {(int(r["x"]), int(r["y"]), int(r["path"])): int(r["kmer_id"]) for r in exp}
        for r in got[j]:
            key = (int(r["x"]), int(r["y"]), int(r["path"]))
            if key in ek:
                assert ek[key] == int(r["kmer_id"]), key
        worst = max(worst, w)
        most_paths = max(most_paths, int(exp["path"].max()) if len(exp) else 0)
    if threshold < 0.05:
        assert most_paths >= 1, "no row of a second path: the case does not exercise several paths per cell"
    print("HDP x CpG ambiguity, threshold %g: worst |d prob_e7| = %d; regions %d (register %d, ring %d, strip %d)"
          % (threshold, worst, st.n_regions, st.n_fast_regions, st.n_ring_regions, st.n_strip_regions))


def test_hdp_with_the_default_ambiguity_table_against_the_oracle(oracle):
    """create_ambig_bases' own table (impl/pairwiseAligner.c:32-65): L = C/E/O (three-way), P = C/E, X = A/C/G/T (all four) --
    also next to each other, so that a cell holds up to 24 paths"""
    pm, om = _models(oracle)
    p = sa.default_params(threshold=0.01)
    op = cases.oracle_params(oracle, p)
    amb_p, amb_o = sa.default_ambig(), oracle.ambig_map()
    jobs = _jobs(pm, 700, 131, n=2)
    jobs[0]["ref"] = _mark(_mark(jobs[0]["ref"], "L", every=2), "P", every=3, what="CA")
    r = list(jobs[1]["ref"])
    for pos in (50, 51, 120, 200, 202):
        r[pos] = "X"
    r[121] = "L"
    jobs[1]["ref"] = "".join(r)
    b = sa.Batch(pm, p, jobs, ambig=amb_p)
    b.run()
    got = [b.pairs(j) for j in range(len(jobs))]
    b.close()
    for j, job in enumerate(jobs):
        exp = cases.oracle_pairs(oracle, om, job, op, ambig=amb_o)
        assert int(exp["path"].max()) >= 2
        w, lonely = cases.compare_pairs(got[j], exp, TOL_E7, p.threshold)
        assert lonely <= max(4, len(exp) // 500)
        assert cases.same_order(got[j], exp)


def test_hdp_ambiguity_is_independent_of_the_kernel_family(oracle, monkeypatch):
    """the same jobs through the default routing and through the memory-resident kernels (SA_FLAG_FORCE_GENERIC, the checker
    of round 3): same rows, posteriors within 2e-7 of each other"""
    pm, om = _models(oracle)
    p = sa.default_params(threshold=0.01)
    amb_p = sa.default_ambig({"X": "CE"})
    jobs = _jobs(pm, 900, 231, n=3)
    for j in jobs:
        j["ref"] = _mark(j["ref"], "X")
    jobs.append(_thin(jobs[0], 37))
    outs = []
    for flags in (0, sa.FLAG_FORCE_GENERIC):
        b = sa.Batch(pm, p, jobs, ambig=amb_p, flags=flags)
        b.run()
        outs.append([b.pairs(j) for j in range(len(jobs))])
        b.close()
    for a, c in zip(*outs):
        cases.compare_pairs(a, c, 2, p.threshold)
        assert cases.same_order(a, c)


def test_hdp_wide_bands_on_the_strip_kernels(oracle, monkeypatch):
    """HDP emissions with the anchors a real guide alignment leaves (bands of 100-300 cells, one path per cell): the strip kernels
    read the emission plane; against the oracle, and bit-identical to the ring kernels' one-path HDP flavour (SA_STRIP=0)."""
    pm, om = _models(oracle)
    p = sa.default_params(threshold=0.05)
    op = cases.oracle_params(oracle, p)
    jobs = [_thin(j, 29) for j in _jobs(pm, 2600, 431, n=3)] + [_thin(_jobs(pm, 300, 461, n=1)[0], 1000)]
    b = sa.Batch(pm, p, jobs)
    b.run()
    st = b.stats()
    got = [b.pairs(j) for j in range(len(jobs))]
    b.close()
    assert st.n_strip_regions >= 3, (st.n_regions, st.n_fast_regions, st.n_ring_regions, st.n_strip_regions)
    for j, job in enumerate(jobs):
        exp = cases.oracle_pairs(oracle, om, job, op)
        assert len(exp) > 0.05 * len(job["events"])
        w, lonely = cases.compare_pairs(got[j], exp, TOL_E7, p.threshold)
        assert lonely <= max(4, len(exp) // 500)
        assert cases.same_order(got[j], exp)
    monkeypatch.setenv("SA_STRIP", "0")
    b = sa.Batch(pm, p, jobs)
    b.run()
    st2 = b.stats()
    ring = [b.pairs(j) for j in range(len(jobs))]
    b.close()
    assert st2.n_strip_regions == 0 and st2.n_ring_regions == st.n_ring_regions
    for a, c in zip(got, ring):
        assert np.array_equal(a, c)


def _write_fasta(path, name, seq, width=60):
    with open(path, "w") as f:
        f.write(">%s\n" % name)
        for i in range(0, len(seq), width):
            f.write(seq[i:i + width] + "\n")
    with open(path + ".fai", "w") as f:
        f.write("%s\t%d\t%d\t%d\t%d\n" % (name, len(seq), len(name) + 2, width, width + 1))


def test_signalmachine_sm3hdp_variant_caller_output(oracle, tmp_path):
    """`signalMachine --sm3Hdp -v <.nhdp> -s 1` on the bundled 2-D read's template strand with X at four reference positions
    (default table: X = A/C/G/T): every row of the variant-caller TSV against the CPU restatement's pairs."""
    npread_path = os.path.join(cases.GOLDEN, "npReads", "ZymoC_ch_1_file1.npRead")
    r = oracle.parse_npread(npread_path)
    read = r["template_read"]
    L = len(read) - 12
    ref = list(read[:L])
    xs = (60, 61, 140, 200)
    for pos in xs:
        ref[pos] = "X"
    ref = "".join(ref)
    fasta = str(tmp_path / "ref.fa")
    _write_fasta(fasta, "chrV", ref + "ACGTACGTAC")
    cigar = str(tmp_path / "guide.cigar")
    with open(cigar, "w") as f:
        f.write("cigar: rv 0 %d + chrV 0 %d + 1 M %d\n" % (L, L, L))
    out = str(tmp_path / "vc.tsv")
    pr = subprocess.run([BIN, "--sm3Hdp", "-T", cases.MODEL_R73, "-v", cases.NHDP, "-q", npread_path, "-f", fasta, "-n", "chrV",
                         "-p", cigar, "-u", out, "-L", "rv", "-s", "1", "-g", "100", "-D", "0.02"],
                        capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr
    assert "SUCCESS" in pr.stderr
    om = oracle.Model.from_file(cases.MODEL_R73)
    ev = r["template_events"].copy()
    est = oracle.estimate_params(om, r["template_strand_event_map"], ev, read)   # before the HDP means are installed
    om.load_hdp(cases.NHDP)
    om.set_to_hdp_expected_values()
    em = r["template_strand_event_map"]
    gx, gy = oracle.guide_to_anchors(0, L, 1, 0, [(0, L)], 14)
    ax, ay = oracle.remap_anchors(gx, gy, em, 0)
    lo, hi = int(em[0]), int(em[L - 1])
    om.set_read_params(est["scale"], est["shift"], est["var"])
    pairs = oracle.align(om, ref, ev[lo:hi], ax, ay, oracle.default_params(threshold=0.02), ambig=oracle.ambig_map())
    k, alpha = om.k, om.alphabet
    exp = {}
    for p in pairs:
        x = int(p["x"])
        window = ref[x:x + k]
        if "X" not in window:
            continue
        kid, kmer = int(p["kmer_id"]), ""
        for _ in range(k):
            kmer = alpha[kid % len(alpha)] + kmer
            kid //= len(alpha)
        for q in range(k):
            if window[q] == "X":
                exp.setdefault((int(p["y"]) + lo, x + q, kmer[q]), []).append(int(p["prob_e7"]) / 1e7)
    rows = [l.rstrip("\n").split("\t") for l in open(out)]
    got = {}
    for g in rows:
        assert len(g) == 9 and g[4] == "t" and g[5] == "forward" and g[6] == "rv" and g[8] == "chrV" and g[2] in "ACGT"
        got.setdefault((int(g[0]), int(g[1]), g[2]), []).append(float(g[3]))
    assert {pos for (_, pos, _) in got} <= set(xs) and len(got) > 20
    assert len({b for (_, _, b) in got}) >= 2, "one base only: the rows do not show several paths"
    lonely = 0
    for key in set(exp) | set(got):
        e, g = sorted(exp.get(key, [])), sorted(got.get(key, []))
        if len(e) == len(g):
            assert all(abs(a - c) <= 1e-5 + 5e-7 for a, c in zip(e, g)), (key, e, g)
        else:   # only rows sitting on the threshold may be on one side only
            assert all(abs(v - 0.02) <= 2e-5 for v in (e + g)), (key, e, g)
            lonely += 1
    assert lonely <= 2
