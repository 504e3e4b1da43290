"""End-to-end test of the signalMachine drop-in: argv as src/signalalign/signalAlignment.py:450-463 builds it,
real input files, TSV / stdout / stderr checked against the oracle plus the reference's format strings."""
import os
import subprocess

import numpy as np
import pytest

import format_checks as fc
import sa_cases as cases

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "signalalign_amd", "bin", "signalMachine")


def _write_fasta(path, name, seq, width=60):
    with open(path, "w") as f:
        f.write(">%s\n" % name)
        for i in range(0, len(seq), width):
            f.write(seq[i:i + width] + "\n")
    with open(path + ".fai", "w") as f:
        f.write("%s\t%d\t%d\t%d\t%d\n" % (name, len(seq), len(name) + 2, width, width + 1))


def _expected_full_rows(oracle, model_path, npread_path, ref_name, label, ref_start, read_start, L, params):
    """What writePosteriorProbsFull (impl/signalMachine.c:89-159) prints for a forward template alignment."""
    r = oracle.parse_npread(npread_path)
    om = oracle.Model.from_file(model_path)
    ev = r["template_events"].copy()
    read = r["template_read"]
    pr = oracle.estimate_params(om, r["template_strand_event_map"], ev, read)
    target = read[read_start:read_start + L]           # the reference IS the read here
    gx, gy = oracle.guide_to_anchors(ref_start, ref_start + L, 1, read_start, [(0, L)], 14)
    em = r["template_strand_event_map"]
    ax, ay = oracle.remap_anchors(gx, gy, em, read_start)
    lo, hi = int(em[read_start]), int(em[read_start + L - 1])
    om.set_read_params(pr["scale"], pr["shift"], pr["var"])
    pairs = oracle.align(om, target, ev[lo:hi], ax, ay, params)
    tab = om.match_table()
    k = om.k
    alpha = om.alphabet
    rows = []
    for p in pairs:
        x, y = int(p["x"]), int(p["y"]) + lo
        kid = int(p["kmer_id"])
        kmer = ""
        t = kid
        for _ in range(k):
            kmer = alpha[t % len(alpha)] + kmer
            t //= len(alpha)
        e_mean, e_noise = tab[5 * kid], tab[5 * kid + 2]
        desc = (ev[y, 0] + pr["var"] * e_mean - pr["scale"] * e_mean - pr["shift"]) / pr["var"]
        rows.append("%s\t%d\t%s\t%s\t%s\t%d\t%f\t%f\t%f\t%s\t%f\t%f\t%f\t%f\t%f\t%s\n" % (
            ref_name, x + ref_start, target[x:x + k], label, "t", y, ev[y, 0], ev[y, 1], ev[y, 2], target[x:x + k],
            e_mean * pr["scale"] + pr["shift"], e_noise * pr["scale_sd"], int(p["prob_e7"]) / 1e7, desc, e_mean, kmer))
    score = 100.0 * float(pairs["prob_e7"].astype(np.float64).sum()) / (len(pairs) * 1e7)
    return rows, len(gx), len(pairs), score


@pytest.mark.parametrize("npread,model", [("c2925_ecoli_ch34_read1023.npRead", cases.MODEL_5MER),
                                          ("r9p4_oneD.npRead", cases.MODEL_6MER)])
def test_signalmachine_full_tsv(oracle, tmp_path, npread, model):
    assert os.path.exists(BIN), "signalMachine is not built"
    npread_path = os.path.join(cases.GOLDEN, "npReads", npread)
    r = oracle.parse_npread(npread_path)
    read = r["template_read"]
    # reference contig = 200 unrelated bases + the read + 150 bases; the guide alignment skips 10 read bases
    rng = np.random.default_rng(1)
    pre = "".join("ACGT"[i] for i in rng.integers(0, 4, 200))
    post = "".join("ACGT"[i] for i in rng.integers(0, 4, 150))
    read_start, L = 10, len(read) - 25
    contig = pre + read[read_start:] + post
    ref_start = len(pre)
    fasta = str(tmp_path / "ref.fa")
    _write_fasta(fasta, "chrTest", contig)
    cigar = str(tmp_path / "guide.cigar")
    with open(cigar, "w") as f:
        f.write("cigar: read1 %d %d + chrTest %d %d + 1 M %d\n" % (read_start, read_start + L, ref_start, ref_start + L, L))
    out = str(tmp_path / "out.tsv")
    argv = [BIN, "--sm3Hdp"][:1] + ["-T", model, "-q", npread_path, "-f", fasta, "-n", "chrTest", "-p", cigar, "-u", out,
                                    "-L", "read1", "-x", "50", "-D", "0.01", "-m", "14", "-g", "100", "-s", "0"]
    pr = subprocess.run(argv, capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr
    assert "SUCCESS" in pr.stderr  # what signalAlignment.py:480 keys on
    assert "signalAlign - SUCCESS: finished alignment of query read1, exiting" in pr.stderr
    params = oracle.default_params()
    rows, n_anchors, n_pairs, score = _expected_full_rows(oracle, model, npread_path, "chrTest", "read1", ref_start,
                                                          read_start, L, params)
    got = open(out).readlines()
    assert len(got) == len(rows)
    # the checker that accepts the reference's own golden full TSV (tests/test_format_fixtures.py) accepts this file
    fc.check_full_rows(open(out).read(), 5 if model == cases.MODEL_5MER else 6, "ACGT")
    bad = 0
    for g, e in zip(got, rows):
        if g != e:
            gf, ef = g.rstrip("\n").split("\t"), e.rstrip("\n").split("\t")
            # only the posterior column may differ, and by no more than 1e-5
            assert gf[:12] == ef[:12] and gf[13:] == ef[13:], (g, e)
            assert abs(float(gf[12]) - float(ef[12])) <= 1e-5
            bad += 1
    assert bad <= len(rows) // 100
    # stdout summary line: "<label> <nAnchors>\t<nPairs>(<score %f>)\t\n"
    head = pr.stdout.strip("\n").split("\t")
    assert head[0] == "read1 %d" % n_anchors
    assert head[1].startswith("%d(" % n_pairs)
    assert abs(float(head[1][head[1].index("(") + 1:-1]) - score) < 1e-3
    # round 6: -s 0 on a reference without ambiguity letters comes over PCIe as 8-byte records (SA_FLAG_PAIRS8: the k-mer of a pair is
    # the reference's at x); with 16-byte records (SA_CLI_PAIRS16=1) the file is the same, byte for byte
    out16 = str(tmp_path / "out16.tsv")
    argv16 = [out16 if a == out else a for a in argv]
    pr16 = subprocess.run(argv16, capture_output=True, text=True, timeout=300, env=dict(os.environ, SA_CLI_PAIRS16="1"))
    assert pr16.returncode == 0 and open(out16, "rb").read() == open(out, "rb").read() and pr16.stdout == pr.stdout
    # appending, not truncating (fopen "a")
    pr = subprocess.run(argv, capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0 and len(open(out).readlines()) == 2 * len(rows)


def test_signalmachine_errors_like_the_reference(tmp_path):
    pr = subprocess.run([BIN], capture_output=True, text=True)
    assert pr.returncode != 0 and "Missing model files" in pr.stderr
    pr = subprocess.run([BIN, "-T", cases.MODEL_5MER], capture_output=True, text=True)
    assert pr.returncode != 0 and "Need to provide input guide alignments" in pr.stderr
    pr = subprocess.run([BIN, "--help"], capture_output=True, text=True)
    assert pr.returncode == 1 and "signalMachine - Align ONT ionic current" in pr.stderr


def test_signalmachine_variant_caller_output(oracle, tmp_path):
    # -s 1: only k-mers holding X are reported, one row per X position (impl/signalMachine.c:161-232)
    npread_path = os.path.join(cases.GOLDEN, "npReads", "c2925_ecoli_ch34_read1023.npRead")
    r = oracle.parse_npread(npread_path)
    read = r["template_read"]
    L = len(read) - 12
    ref = list(read[:L])
    for pos in (60, 61, 140, 200):
        ref[pos] = "X"
    ref = "".join(ref)
    fasta = str(tmp_path / "ref.fa")
    _write_fasta(fasta, "chrX", ref + "ACGTACGTAC")
    cigar = str(tmp_path / "guide.cigar")
    with open(cigar, "w") as f:
        f.write("cigar: r 0 %d + chrX 0 %d + 1 M %d\n" % (L, L, L))
    out = str(tmp_path / "vc.tsv")
    pr = subprocess.run([BIN, "-T", cases.MODEL_5MER, "-q", npread_path, "-f", fasta, "-n", "chrX", "-p", cigar, "-u", out,
                         "-L", "r", "-s", "1", "-g", "100"], capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr
    rows = [l.rstrip("\n").split("\t") for l in open(out)]
    assert rows, "no variant rows"
    positions = {int(r[1]) for r in rows}
    assert positions <= {60, 61, 140, 200} and {60, 140, 200} <= positions
    assert all(r[2] in "ACGT" and r[4] == "t" and r[5] == "forward" and r[6] == "r" and r[8] == "chrX" for r in rows)
    # per (event, position) the called bases' posteriors are probabilities
    for r in rows:
        assert 0.01 <= float(r[3]) <= 1.0


def test_variant_caller_rows_are_filtered_on_the_device(oracle, tmp_path):
    """-s 1 keeps only the rows whose reference k-mer holds an X (writePosteriorProbsVC, impl/signalMachine.c:161-232).  The library
    drops the others on the device (SA_FLAG_VC_ROWS) and keeps their count and probability sum for the summary line: the file and
    the summary line are byte-identical to the run that fetches every pair and filters while it writes (SA_CLI_VC_ON_HOST=1) -- a 1-D
    read and both strands of the bundled 2-D read --, and the library call returns exactly the filtered subset of the unfiltered call -- device finalisation and SA_FLAG_EXACT's host
    finalisation alike."""
    npread_path = os.path.join(cases.GOLDEN, "npReads", "r9p4_oneD.npRead")
    r = oracle.parse_npread(npread_path)
    read = r["template_read"]
    L = 1500
    ref = list(read[:L])
    cpg = [i for i in range(100, L - 100) if read[i:i + 2] == "CG"][:20]
    for pos in cpg:
        ref[pos] = "X"
    ref = "".join(ref)
    fasta = str(tmp_path / "ref.fa")
    _write_fasta(fasta, "chrA", ref + "ACGTACGTAC")
    cigar = str(tmp_path / "guide.cigar")
    with open(cigar, "w") as f:
        f.write("cigar: r 0 %d + chrA 0 %d + 1 M %d\n" % (L, L, L))
    amb = str(tmp_path / "ce.positions")
    with open(amb, "w") as f:
        f.write("X\tCE\n")
    base = [BIN, "-T", cases.MODEL_CPG, "-q", npread_path, "-f", fasta, "-n", "chrA", "-p", cigar, "-L", "r", "-s", "1", "-g", "100", "-a", amb]
    outs = {}
    for name, env in (("device", {}), ("host", {"SA_CLI_VC_ON_HOST": "1"})):
        out = str(tmp_path / (name + ".tsv"))
        pr = subprocess.run(base + ["-u", out], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert pr.returncode == 0, pr.stderr
        outs[name] = (open(out).read(), pr.stdout)
    assert outs["device"][0] and outs["device"] == outs["host"]
    n_all = int(outs["device"][1].split("\t")[1].split("(")[0])
    assert n_all > 10 * len(outs["device"][0].splitlines()) // 6        # the summary line still counts every pair
    # both strands of the bundled 2-D read (--twoD: two batches, the template's released before the complement's runs), the
    # reference's own contig with an X at fifteen cytosines of the aligned window
    import json
    cig = json.load(open(os.path.join(cases.GOLDEN, "cigars", "zymoC_lastz_anchors.json")))["calls"][0]["cigars"][0].split()
    cigar2 = str(tmp_path / "guide2d.cigar")
    with open(cigar2, "w") as f:
        f.write(" ".join(["cigar:", "read2d"] + cig[2:5] + ["ZYMO"] + cig[6:]) + "\n")
    zymo = "".join(l.strip() for l in open(os.path.join(cases.GOLDEN, "sequences", "zymo_sequence.fasta")) if not l.startswith(">"))
    t0, t1 = sorted((int(cig[6]), int(cig[7])))
    z = list(zymo)
    cs = [i for i in range(t0 + 20, t1 - 20) if z[i] == "C"]
    for i in cs[::max(1, len(cs) // 15)][:15]:
        z[i] = "X"
    fasta2 = str(tmp_path / "zymo_x.fa")
    _write_fasta(fasta2, "ZYMO", "".join(z))
    model_c = os.path.join(cases.GOLDEN, "models", "testModelR73_acegot_complement.model")
    npread2 = os.path.join(cases.GOLDEN, "npReads", "ZymoC_ch_1_file1.npRead")
    outs2 = {}
    for name, env in (("device", {}), ("host", {"SA_CLI_VC_ON_HOST": "1"})):
        out = str(tmp_path / (name + "_2d.tsv"))
        pr = subprocess.run([BIN, "-T", cases.MODEL_R73, "-C", model_c, "-q", npread2, "-f", fasta2, "-n", "ZYMO", "-p", cigar2, "-u", out,
                             "-L", "read2d", "--twoD", "-s", "1", "-g", "100"], capture_output=True, text=True, timeout=300,
                            env=dict(os.environ, **env))
        assert pr.returncode == 0, pr.stderr
        outs2[name] = (open(out).read(), pr.stdout)
    assert outs2["device"] == outs2["host"]
    strands = {l.split("\t")[4] for l in outs2["device"][0].splitlines()}
    assert strands == {"t", "c"}, strands
    # the library call
    import signalalign_amd as sa
    from signalalign_amd import synth
    pm = sa.Model.load(cases.MODEL_CPG)
    kk = synth.parse_model_table(cases.MODEL_CPG)[1]
    jobs = cases.synthetic_jobs(cases.MODEL_CPG, 6, 900, 4242)
    rng = np.random.default_rng(3)
    for j in jobs:                                    # X at a few cytosines of every read but the last (no X: every row goes)
        s = list(j["ref"])
        cs = [i for i in range(len(s) - 1) if s[i] == "C"]
        if j is not jobs[-1]:
            for i in rng.choice(cs, size=min(12, len(cs)), replace=False):
                s[int(i)] = "X"
        j["ref"] = "".join(s)
    ambig = sa.default_ambig({"X": "CE"})
    p = sa.default_params(threshold=0.01)
    for flags in (0, sa.FLAG_EXACT):
        full = sa.Batch(pm, p, jobs, ambig=ambig, flags=flags)
        full.run()
        vc = sa.Batch(pm, p, jobs, ambig=ambig, flags=flags | sa.FLAG_VC_ROWS)
        vc.run()
        for j, job in enumerate(jobs):
            a, b = full.pairs(j), vc.pairs(j)
            has_x = np.array([("X" in job["ref"][x:x + kk]) for x in a["x"]], dtype=bool)
            assert np.array_equal(a[has_x], b), (flags, j)
            assert vc.all_pairs_summary(j) == (len(a), int(a["prob_e7"].sum())) == full.all_pairs_summary(j)
        assert len(vc.pairs(len(jobs) - 1)) == 0 and vc.n_pairs(0) > 0
        if flags == 0:   # a filtered batch holds the X rows only: the MEA path over all posteriors is refused, not computed on them
            with pytest.raises(sa.SaError) as ei:
                vc.mea()
            assert ei.value.code == -7
        full.close(); vc.close()
    with pytest.raises(sa.SaError) as ei:       # 8-byte records do not name the k-mer the filter looks at
        sa.Batch(pm, p, jobs, ambig=ambig, flags=sa.FLAG_VC_ROWS | sa.FLAG_PAIRS8)
    assert ei.value.code == -1


def test_signalmachine_ambig_model_file(oracle, tmp_path):
    """-a <file> (create_ambig_bases2, impl/pairwiseAligner.c:68-92; impl/signalMachine.c:649-655): the file REPLACES the
    built-in ambiguity table.  With `X<TAB>AT` only A or T can be called at the X positions (which really hold a C: the
    built-in table, X -> ACGT, calls C there); with `X<TAB>CE` over the CpG model only cytosine / 5-methylcytosine.  The
    reference's own fixture
    (tests/test_position_code/test_positions_encoding.positions) loads as well: it does not name X, so X is then a letter
    outside the alphabet and the run fails as the reference's kmer_id does."""
    npread_path = os.path.join(cases.GOLDEN, "npReads", "r9p4_oneD.npRead")
    r = oracle.parse_npread(npread_path)
    read = r["template_read"]
    L = 1500
    ref = list(read[:L])
    cpg = [i for i in range(100, L - 100) if read[i:i + 2] == "CG"][:12]
    for pos in cpg:
        ref[pos] = "X"
    fasta = str(tmp_path / "ref.fa")
    _write_fasta(fasta, "chrA", "".join(ref) + "ACGTACGTAC")
    cigar = str(tmp_path / "guide.cigar")
    with open(cigar, "w") as f:
        f.write("cigar: r 0 %d + chrA 0 %d + 1 M %d\n" % (L, L, L))
    amb = str(tmp_path / "ce.positions")
    with open(amb, "w") as f:
        f.write("X\tCE\n")
    amb_at = str(tmp_path / "at.positions")
    with open(amb_at, "w") as f:
        f.write("X\tAT\nR\tAG\n")
    base = [BIN, "-T", cases.MODEL_CPG, "-q", npread_path, "-f", fasta, "-n", "chrA", "-p", cigar, "-L", "r", "-s", "1", "-g", "100"]

    def rows_of(extra, name):
        out = str(tmp_path / name)
        pr = subprocess.run(base + ["-u", out] + extra, capture_output=True, text=True, timeout=300)
        assert pr.returncode == 0, pr.stderr
        return [l.rstrip("\n").split("\t") for l in open(out)]
    with_file = rows_of(["-a", amb], "ce.tsv")
    builtin = rows_of([], "acgt.tsv")
    assert with_file and {int(r[1]) for r in with_file} <= set(cpg)
    assert {r[2] for r in with_file} <= {"C", "E"} and "C" in {r[2] for r in with_file}
    assert {r[2] for r in builtin} <= set("ACGT") and "C" in {r[2] for r in builtin}
    only_at = rows_of(["-a", amb_at], "at.tsv")
    assert only_at and {r[2] for r in only_at} <= {"A", "T"} and {int(r[1]) for r in only_at} <= set(cpg)
    # the same answer as the library called with the same table
    ref_fixture = os.path.join(cases.GOLDEN, "position_code", "test_positions_encoding.positions")
    pr = subprocess.run(base + ["-u", str(tmp_path / "bad.tsv"), "-a", ref_fixture], capture_output=True, text=True, timeout=300)
    assert pr.returncode != 0
    pr = subprocess.run(base + ["-u", str(tmp_path / "bad.tsv"), "-a", str(tmp_path / "nope")], capture_output=True, text=True,
                        timeout=300)
    assert pr.returncode != 0 and "Couldn't open" in pr.stdout


def test_signalmachine_rna(oracle, tmp_path):
    """--rna (impl/signalMachine.c:716-724; impl/fasta_handler.c:47-102; writers :145-148, :166-170) on the reference's own
    fake_rna FASTA pair (tests/test_sequences/fake_rna_replace/{forward,backward}.fake_rna_atg.fake_rna_ref.fa, every A of
    an ATG replaced by X on either strand; committed as data).  RNA is read 3' -> 5': the guide alignment's read interval is
    mirrored, its operations reversed, the template target becomes the REVERSED forward reference and the strand flips.
    The read is synthetic (the reference ships no RNA .npRead): its sequence is that reversed target with the true base at
    the X positions.  Expected rows come from the CPU restatement run on inputs transformed here by the reference's rules."""
    model = cases.MODEL_6MER
    seqdir = os.path.join(cases.GOLDEN, "sequences")
    fwd_fa = os.path.join(seqdir, "fake_rna_replace", "forward.fake_rna_atg.fake_rna_ref.fa")
    bwd_fa = os.path.join(seqdir, "fake_rna_replace", "backward.fake_rna_atg.fake_rna_ref.fa")

    def seq_of(path):
        return "".join(open(path).read().split("\n")[1:])
    F, F0 = seq_of(fwd_fa), seq_of(os.path.join(seqdir, "fake_rna_ref.fa"))
    a, b = 30, 1000
    L = b - a
    target = F[a:b][::-1]                      # what referenceSequence_getTemplateTarget returns under --rna, '+' cigar
    pad = 8
    rng = np.random.default_rng(5)
    read = "".join("ACGT"[i] for i in rng.integers(0, 4, pad)) + F0[a:b][::-1] + "".join("ACGT"[i] for i in rng.integers(0, 4, pad))
    ev, emap = cases.events_for_sequence(read, model, 77)
    npread_path = str(tmp_path / "rna.npRead")
    cases.write_npread_1d(npread_path, read, emap, ev)
    Lr = len(read)
    cigar = str(tmp_path / "guide.cigar")
    with open(cigar, "w") as f:     # read interval [pad, Lr - pad) mirrors onto itself: Lr - (Lr - pad) = pad
        f.write("cigar: rna1 %d %d + rna_fake %d %d + 1 M %d\n" % (pad, Lr - pad, a, b, L))
    out = str(tmp_path / "rna.tsv")
    pr = subprocess.run([BIN, "-T", model, "-q", npread_path, "-f", fwd_fa, "-b", bwd_fa, "-n", "rna_fake", "-p", cigar, "-u", out,
                         "-L", "rna1", "-s", "0", "-g", "100", "--rna"], capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr
    assert "SUCCESS" in pr.stderr
    # ---- the same through the CPU restatement ----
    r = oracle.parse_npread(npread_path)
    om = oracle.Model.from_file(model)
    ev2 = r["template_events"].copy()
    prm = oracle.estimate_params(om, r["template_strand_event_map"], ev2, read)
    start2, end2 = Lr - (Lr - pad), Lr - pad                     # :716-720
    start1, end1, strand1 = b, a, 0                              # fasta_handler.c:83-90: swapped, strand flipped
    gx, gy = oracle.guide_to_anchors(start1, end1, strand1, start2, [(0, L)], 14)
    em = r["template_strand_event_map"]
    ax, ay = oracle.remap_anchors(gx, gy, em, start2)
    lo, hi = int(em[start2]), int(em[end2 - 1])
    om.set_read_params(prm["scale"], prm["shift"], prm["var"])
    pairs = oracle.align(om, target, ev2[lo:hi], ax, ay, oracle.default_params())
    assert pairs["path"].max() >= 1                              # X positions really are ambiguous (ACGT)
    tab, k, alpha = om.match_table(), om.k, om.alphabet
    ref_len = len(target)
    rows = []
    for p in pairs:
        x, y, kid = int(p["x"]), int(p["y"]) + lo, int(p["kmer_id"])
        kmer, t = "", kid
        for _ in range(k):
            kmer = alpha[t % len(alpha)] + kmer
            t //= len(alpha)
        x_adj = (ref_len - k) - (x + (ref_len - start1))         # adjustReferenceCoordinate, template strand mapped backward
        k_i = target[x:x + k]                                    # reference k-mer: reverse complement twice (:64-69, :143-146)
        e_mean, e_noise = tab[5 * kid], tab[5 * kid + 2]
        desc = (ev2[y, 0] + prm["var"] * e_mean - prm["scale"] * e_mean - prm["shift"]) / prm["var"]
        rows.append("%s\t%d\t%s\t%s\t%s\t%d\t%f\t%f\t%f\t%s\t%f\t%f\t%f\t%f\t%f\t%s\n" % (
            "rna_fake", x_adj, k_i, "rna1", "t", y, ev2[y, 0], ev2[y, 1], ev2[y, 2], k_i,
            e_mean * prm["scale"] + prm["shift"], e_noise * prm["scale_sd"], int(p["prob_e7"]) / 1e7, desc, e_mean, kmer))
    got = open(out).readlines()
    assert len(got) == len(rows) and len(rows) > 1000
    bad = 0
    for g, e in zip(got, rows):
        if g != e:
            gf, ef = g.rstrip("\n").split("\t"), e.rstrip("\n").split("\t")
            assert gf[:12] == ef[:12] and gf[13:] == ef[13:], (g, e)
            assert abs(float(gf[12]) - float(ef[12])) <= 1e-5
            bad += 1
    assert bad <= len(rows) // 50
    # variant-calling output: under --rna the strand label flips (:166-170): a '+' guide alignment is reported "backward"
    vc = str(tmp_path / "rna_vc.tsv")
    pr = subprocess.run([BIN, "-T", model, "-q", npread_path, "-f", fwd_fa, "-b", bwd_fa, "-n", "rna_fake", "-p", cigar, "-u", vc,
                         "-L", "rna1", "-s", "1", "-g", "100", "--rna"], capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr
    vrows = [l.rstrip("\n").split("\t") for l in open(vc)]
    assert vrows and all(v[4] == "t" and v[5] == "forward" for v in vrows)

    # ---- the same read family mapped to the MINUS strand.  The reference's own RNA outputs fix the conventions
    # (tests/test_variantCalled_files/rna/: in 8898d755-...sm.backward.tsv positions grow with the events, the target k-mer is the
    # COMPLEMENT of the forward reference at its position -- not reversed -- and the reference k-mer column its reverse
    # complement; in 7d31de25-...sm.forward.tsv positions fall, target k-mer = REVERSED forward reference = reference k-mer column)
    B = seq_of(bwd_fa)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    read_m = ("".join("ACGT"[i] for i in rng.integers(0, 4, pad)) + "".join(comp[c] for c in F0[a:b]) +
              "".join("ACGT"[i] for i in rng.integers(0, 4, pad)))
    ev_m, emap_m = cases.events_for_sequence(read_m, model, 78)
    np_m = str(tmp_path / "rna_minus.npRead")
    cases.write_npread_1d(np_m, read_m, emap_m, ev_m)
    cigar_m = str(tmp_path / "guide_minus.cigar")
    with open(cigar_m, "w") as f:
        f.write("cigar: rna2 %d %d + rna_fake %d %d - 1 M %d\n" % (pad, Lr - pad, b, a, L))
    out_m = str(tmp_path / "rna_minus.tsv")
    pr = subprocess.run([BIN, "-T", model, "-q", np_m, "-f", fwd_fa, "-b", bwd_fa, "-n", "rna_fake", "-p", cigar_m, "-u", out_m,
                         "-L", "rna2", "-s", "0", "-g", "100", "--rna"], capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr
    r2 = oracle.parse_npread(np_m)
    ev3 = r2["template_events"].copy()
    prm2 = oracle.estimate_params(om, r2["template_strand_event_map"], ev3, read_m)
    target_m = B[a:b]                                           # the backward file's bases in forward order
    gx, gy = oracle.guide_to_anchors(a, b, 1, pad, [(0, L)], 14)   # start1 and end1 swapped, strand flipped to forward
    em2 = r2["template_strand_event_map"]
    ax, ay = oracle.remap_anchors(gx, gy, em2, pad)
    lo2, hi2 = int(em2[pad]), int(em2[Lr - pad - 1])
    om.set_read_params(prm2["scale"], prm2["shift"], prm2["var"])
    pairs_m = oracle.align(om, target_m, ev3[lo2:hi2], ax, ay, oracle.default_params())
    got_m = [l.rstrip("\n").split("\t") for l in open(out_m)]
    assert len(got_m) == len(pairs_m) > 1000
    rc_ = lambda s_: "".join({"A": "T", "C": "G", "G": "C", "T": "A"}.get(c, c) for c in reversed(s_))
    for g, p_ in zip(got_m, pairs_m):
        x, y = int(p_["x"]), int(p_["y"]) + lo2
        k_i = target_m[x:x + k]
        assert int(g[1]) == x + a and int(g[5]) == y and g[9] == k_i and g[2] == rc_(k_i), (g, x, y)
        assert abs(float(g[12]) - int(p_["prob_e7"]) / 1e7) <= 1e-5
    # the conventions of the reference's files, on positions without an ambiguity letter: complement of the forward
    # reference in place (minus strand), reversed forward reference (plus strand, first run above)
    for g in got_m[::37]:
        if "X" not in g[9]:
            assert g[9] == "".join(comp[c] for c in F0[int(g[1]):int(g[1]) + k])
    for g in got[::37]:
        gf = g.rstrip("\n").split("\t")
        if "X" not in gf[9]:
            assert gf[9] == F0[int(gf[1]):int(gf[1]) + k][::-1] and gf[2] == gf[9]
    vc_m = str(tmp_path / "rna_minus_vc.tsv")
    pr = subprocess.run([BIN, "-T", model, "-q", np_m, "-f", fwd_fa, "-b", bwd_fa, "-n", "rna_fake", "-p", cigar_m, "-u", vc_m,
                         "-L", "rna2", "-s", "1", "-g", "100", "--rna"], capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr
    vm = [l.rstrip("\n").split("\t") for l in open(vc_m)]
    assert vm and all(v[4] == "t" and v[5] == "backward" for v in vm)


def test_signalmachine_two_d_read_against_the_reference_output(oracle, tmp_path):
    """The reference's own output for the bundled 2-D read (tests/test_alignments/zymo_C_test_alignments_sm3/...7f22f937...,
    600 rows committed under tests/golden/format/) against this signalMachine run on the same read (--twoD, template and
    complement models, ZYMO contig, guide alignment from the reference's lastz): rows are matched by (strand, reference
    position, event, k-mer) and the seven columns that do not depend on the HMM's parameters -- event mean after the drift
    correction, noise, duration, scaled model mean and noise, descaled event mean, model mean -- must be the reference's bytes.
    (The posterior column cannot be: the golden file was written with model parameters the tree does not hold; see
    tests/test_host_golden_columns.py.)  At least 80 % of the golden rows must have a partner."""
    import json
    gold = [l.rstrip("\n").split("\t") for l in open(os.path.join(cases.GOLDEN, "format", "zymo_C_sm3_7f22f937.forward.t300_c300.tsv"))]
    cig = json.load(open(os.path.join(cases.GOLDEN, "cigars", "zymoC_lastz_anchors.json")))["calls"][0]["cigars"][0].split()
    # lastz writes "cigar: query s e + target s e + score ops": the same field order signalMachine reads (read first)
    cigar = str(tmp_path / "guide.cigar")
    with open(cigar, "w") as f:
        f.write(" ".join(["cigar:", "read2d"] + cig[2:5] + ["ZYMO"] + cig[6:]) + "\n")
    fasta = os.path.join(cases.GOLDEN, "sequences", "zymo_sequence.fasta")
    out = str(tmp_path / "twod.tsv")
    npread = os.path.join(cases.GOLDEN, "npReads", "ZymoC_ch_1_file1.npRead")
    model_c = os.path.join(cases.GOLDEN, "models", "testModelR73_acegot_complement.model")
    pr = subprocess.run([BIN, "-T", cases.MODEL_R73, "-C", model_c, "-q", npread, "-f", fasta, "-n", "ZYMO", "-p", cigar, "-u", out,
                         "-L", gold[0][3], "--twoD", "-s", "0", "-g", "100"], capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr
    mine = {}
    for l in open(out):
        g = l.rstrip("\n").split("\t")
        mine[(g[4], g[1], g[5], g[15])] = g
    hit = 0
    for g in gold:
        m = mine.get((g[4], g[1], g[5], g[15]))
        if m is None:
            continue
        hit += 1
        assert [m[i] for i in (0, 2, 3, 6, 7, 8, 9, 10, 11, 13, 14)] == [g[i] for i in (0, 2, 3, 6, 7, 8, 9, 10, 11, 13, 14)], (m, g)
    assert hit >= 0.8 * len(gold), hit
    assert any(g[4] == "c" for g in gold) and any(k[0] == "c" for k in mine)


def test_signalmachine_reproduces_the_reference_output_file_of_the_two_d_read(tmp_path):
    """End to end against the reference's binary: its shipped output for the bundled 2-D read (2007 rows, template and complement;
    positions, events and posteriors committed as tests/golden/expected/reference_output_zymo2d.npz) was written with the
    two-distribution emission (tests/test_oracle_reference_outputs.py).  `signalMachine --twoD --emission twoDist` on the same
    read, models and contig must print the same posterior for the same (strand, position, event): median |dp| at the printed
    precision and at least 85 % of the rows within 1e-4 on BOTH strands -- the rest is the guide alignment (bwa's then, lastz's
    here)."""
    import json
    z = np.load(os.path.join(cases.GOLDEN, "expected", "reference_output_zymo2d.npz"))
    cig = json.load(open(os.path.join(cases.GOLDEN, "cigars", "zymoC_lastz_anchors.json")))["calls"][0]["cigars"][0].split()
    cigar = str(tmp_path / "guide.cigar")
    with open(cigar, "w") as f:
        f.write(" ".join(["cigar:", "read2d"] + cig[2:5] + ["ZYMO"] + cig[6:]) + "\n")
    out = str(tmp_path / "twod.tsv")
    pr = subprocess.run([BIN, "-T", cases.MODEL_R73, "-C", os.path.join(cases.GOLDEN, "models", "testModelR73_acegot_complement.model"),
                         "-q", os.path.join(cases.GOLDEN, "npReads", "ZymoC_ch_1_file1.npRead"),
                         "-f", os.path.join(cases.GOLDEN, "sequences", "zymo_sequence.fasta"), "-n", "ZYMO", "-p", cigar, "-u", out,
                         "-L", "read2d", "--twoD", "--emission", "twoDist", "-s", "0", "-g", "100"],
                        capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr
    mine = {}
    for l in open(out):
        g = l.rstrip("\n").split("\t")
        mine[(g[4], int(g[1]), int(g[5]))] = float(g[12])
    for strand in ("t", "c"):
        sel = z["strand"] == strand
        gold = {(strand, int(x), int(y)): float(p) for x, y, p in zip(z["x"][sel], z["y"][sel], z["p"][sel])}
        common = set(mine) & set(gold)
        d = np.array([abs(mine[k_] - gold[k_]) for k_ in common])
        # (measured: template 1031 of 1044 rows, median 3e-7, 91 % within 1e-4; complement 961 of 963, median 0, 89 %)
        assert len(common) >= 0.95 * len(gold) and np.median(d) <= 2e-6 and (d <= 1e-4).mean() >= 0.85, \
            (strand, len(common), len(gold), float(np.median(d)), float((d <= 1e-4).mean()))
        # (round 4: what lies beyond 1e-4 is one factor per checkpoint group -- sa_cases.reference_residual; relative to it the
        # rows agree, both strands)
        rel = cases.reference_residual({k_[1:]: v_ for k_, v_ in mine.items() if k_[0] == strand},
                                       {k_[1:]: v_ for k_, v_ in gold.items()})
        assert rel[2] >= 0.98, (strand, rel[2], rel[4][:5])   # (measured: template 0.995, complement 0.984)
    # the option is not for batches, HDP models or the expectation routine
    pr = subprocess.run([BIN, "-T", cases.MODEL_R73, "--emission", "nope"], capture_output=True, text=True)
    assert pr.returncode != 0 and "--emission takes" in pr.stderr


def test_signalmachine_expectations_file(oracle, tmp_path):
    # -t: the .expectations file of continuousPairHmm_writeToFile (impl/continuousHmm.c:352-408)
    model = cases.MODEL_6MER
    npread_path = os.path.join(cases.GOLDEN, "npReads", "r9p4_oneD.npRead")
    r = oracle.parse_npread(npread_path)
    read = r["template_read"]
    read_start, L = 5, len(read) - 20
    fasta = str(tmp_path / "ref.fa")
    _write_fasta(fasta, "chrE", "ACGT" * 10 + read[read_start:] + "TTTT")
    cigar = str(tmp_path / "guide.cigar")
    with open(cigar, "w") as f:
        f.write("cigar: r %d %d + chrE 40 %d + 1 M %d\n" % (read_start, read_start + L, 40 + L, L))
    out = str(tmp_path / "t.expectations")
    pr = subprocess.run([BIN, "-T", model, "-q", npread_path, "-f", fasta, "-n", "chrE", "-p", cigar, "-t", out,
                         "-L", "r", "-g", "100"], capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr
    assert "signalAlign - writing expectations to file: %s" % out in pr.stderr and "SUCCESS" in pr.stderr
    lines = open(out).read().split("\n")
    om = oracle.Model.from_file(model)
    assert lines[0] == "3\t%d\t%s\t%d\t" % (len(om.alphabet), om.alphabet, om.k)
    # the oracle on the same inputs
    ev = r["template_events"].copy()
    pr_ = oracle.estimate_params(om, r["template_strand_event_map"], ev, read)
    target = read[read_start:read_start + L]
    gx, gy = oracle.guide_to_anchors(40, 40 + L, 1, read_start, [(0, L)], 14)
    em = r["template_strand_event_map"]
    ax, ay = oracle.remap_anchors(gx, gy, em, read_start)
    lo, hi = int(em[read_start]), int(em[read_start + L - 1])
    om.set_read_params(pr_["scale"], pr_["shift"], pr_["var"])
    t, lik, _, _, _ = oracle.expectations(om, target, ev[lo:hi], ax, ay, oracle.default_params())
    f1 = lines[1].split("\t")
    assert len(f1) == 10
    exp = ["%f" % (v + 0.001) for v in t] + ["%f" % lik]
    for g, e in zip(f1, exp):
        assert abs(float(g) - float(e)) <= 2e-6 * max(1.0, abs(float(e))), (f1, exp)
    n_kmers = len(om.alphabet) ** om.k
    tab = om.match_table()
    f2 = lines[2].split("\t")
    assert len(f2) == 5 * n_kmers + 1 and f2[-1] == "" and f2[:5] == ["%f" % v for v in tab[:5]]
    assert lines[3] == "0.000000\t" * (2 * n_kmers)
    assert lines[4] == "0.001000\t" * n_kmers
    assert lines[5] == "0\t" * n_kmers


def test_signalmachine_expectations_file_matches_the_reference_golden_layout(oracle, tmp_path):
    # The reference's golden Gaussian expectations file (tests/test_expectation_files/4f9a316c-...: ACEGT 6-mer) has
    # 6 lines of 4 / 10 / 78125 / 31250 / 15625 / 15625 tokens, all but the transitions line tab-terminated.  The same
    # model (the bundled R9.4 CpG model, ACEGT 6-mer) through -t here must give exactly that layout.
    import gzip
    model = cases.MODEL_CPG
    npread_path = os.path.join(cases.GOLDEN, "npReads", "r9p4_oneD.npRead")
    r = oracle.parse_npread(npread_path)
    read = r["template_read"]
    read_start, L = 0, 1500
    fasta = str(tmp_path / "ref.fa")
    _write_fasta(fasta, "chrE", "ACGT" * 10 + read[:L + 30])
    cigar = str(tmp_path / "guide.cigar")
    with open(cigar, "w") as f:
        f.write("cigar: r %d %d + chrE 40 %d + 1 M %d\n" % (read_start, read_start + L, 40 + L, L))
    out = str(tmp_path / "t.expectations")
    pr = subprocess.run([BIN, "-T", model, "-q", npread_path, "-f", fasta, "-n", "chrE", "-p", cigar, "-t", out,
                         "-L", "r", "-g", "100"], capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr
    ours = fc.check_expectations_file(open(out).read(), 5, "ACEGT", 6)
    gold = fc.check_expectations_file(
        gzip.open(os.path.join(cases.GOLDEN, "format", "4f9a316c-8bb3-410a-8cfc-026061f7e8db.template.expectations.tsv.gz"),
                  "rt").read(), 5, "ACEGT", 6)
    count = lambda l: len([t for t in l.split("\t") if t != ""])
    assert [count(l) for l in ours[:6]] == [count(l) for l in gold[:6]] == [4, 10, 78125, 31250, 15625, 15625]
    assert ours[0] == gold[0]
    # the two dead transitions sit at the pseudocount in both files; the likelihood is the per-diagonal sum (negative)
    to, tg = ours[1].split("\t"), gold[1].split("\t")
    assert to[5] == tg[5] == "0.001000" and to[7] == tg[7] == "0.001000" and float(to[9]) < 0 and float(tg[9]) < 0


def test_signalmachine_batch_front_door(oracle, tmp_path):
    # --batch: several reads in one process / one GPU batch; every read's files and log lines must be what single-read
    # invocations produce
    model = cases.MODEL_6MER
    npread_path = os.path.join(cases.GOLDEN, "npReads", "r9p4_oneD.npRead")
    r = oracle.parse_npread(npread_path)
    read = r["template_read"]
    fasta = str(tmp_path / "ref.fa")
    _write_fasta(fasta, "chrB", "ACGTTGCA" * 20 + read + "GATTACA" * 10)
    specs = [("readA", 0, len(read) - 30), ("readB", 400, 1500), ("readC", 2000, 2500)]
    single_out, manifest_rows = {}, []
    common = ["-T", model, "-f", fasta, "-n", "chrB", "-g", "100", "-x", "50", "-D", "0.01", "-m", "14"]
    for name, start, L in specs:
        cigar = str(tmp_path / (name + ".cigar"))
        with open(cigar, "w") as f:
            f.write("cigar: %s %d %d + chrB %d %d + 1 M %d\n" % (name, start, start + L, 160 + start, 160 + start + L, L))
        out1 = str(tmp_path / (name + ".single.tsv"))
        pr = subprocess.run([BIN] + common + ["-q", npread_path, "-p", cigar, "-u", out1, "-L", name],
                            capture_output=True, text=True, timeout=300)
        assert pr.returncode == 0, pr.stderr
        single_out[name] = (open(out1).read(), pr.stdout)
        manifest_rows.append("\t".join([name, npread_path, cigar, str(tmp_path / (name + ".batch.tsv"))]))
    manifest_rows.insert(1, "# a comment line")
    manifest_rows.append("\t".join(["broken", str(tmp_path / "missing.npRead"), str(tmp_path / "readA.cigar"),
                                    str(tmp_path / "broken.tsv")]))
    # a read whose reference window holds a letter outside the model's alphabet: the planner rejects its job
    # (kmer_id aborts in the reference, impl/nanopore_hdp.c:387-403 -- for that read's process only), so here only that read
    # may fail, not the GPU batch it shares with the others
    contig = "ACGTTGCA" * 20 + read + "GATTACA" * 10
    fasta_n = str(tmp_path / "refN.fa")
    _write_fasta(fasta_n, "chrN", contig[:160 + 700] + "N" + contig[160 + 701:])
    os.remove(fasta_n + ".fai")                  # two records in one file: let the loader scan it (no index)
    with open(fasta, "a") as f:
        f.write(open(fasta_n).read())
    os.remove(fasta + ".fai")
    cigar_n = str(tmp_path / "readN.cigar")
    with open(cigar_n, "w") as f:
        f.write("cigar: readN %d %d + chrN %d %d + 1 M %d\n" % (400, 1900, 560, 2060, 1500))
    manifest_rows.insert(2, "\t".join(["readN", npread_path, cigar_n, str(tmp_path / "readN.tsv"), "-", "chrN"]))
    manifest = str(tmp_path / "manifest.tsv")
    with open(manifest, "w") as f:
        f.write("\n".join(manifest_rows) + "\n")
    pr = subprocess.run([BIN] + common + ["--batch", manifest], capture_output=True, text=True, timeout=600)
    assert pr.returncode == 1  # two reads of the manifest are broken; the others are done
    assert "read broken skipped" in pr.stderr and "3 of 5 reads aligned" in pr.stderr
    assert "read readN skipped: alignment job rejected: k-mer contains a character outside the model alphabet" in pr.stderr
    assert not os.path.exists(str(tmp_path / "readN.tsv"))
    for name, _, _ in specs:
        assert open(str(tmp_path / (name + ".batch.tsv"))).read() == single_out[name][0], name
        assert single_out[name][1] in pr.stdout
        assert "signalAlign - SUCCESS: finished alignment of query %s, exiting" % name in pr.stderr
    assert not os.path.exists(str(tmp_path / "broken.tsv"))
    # the same manifest in slices of two reads (--batch-reads): several GPU batches in one process, the storage of one
    # reused by the next, identical files (appended a second time) and the same summary
    pr = subprocess.run([BIN] + common + ["--batch", manifest, "--batch-reads", "2", "--mea"], capture_output=True, text=True,
                        timeout=600)
    assert pr.returncode == 1 and "3 of 5 reads aligned" in pr.stderr
    for name, _, _ in specs:
        assert open(str(tmp_path / (name + ".batch.tsv"))).read() == 2 * single_out[name][0], name
        assert single_out[name][1] in pr.stdout
        mea_rows = open(str(tmp_path / (name + ".batch.tsv.mea"))).readlines()
        full_rows = set(single_out[name][0].splitlines(True))
        assert len(mea_rows) > 0 and all(row in full_rows for row in mea_rows)
        events = [int(row.split("\t")[5]) for row in mea_rows]
        assert events == sorted(set(events))               # one row per event, ascending


def _revcomp(s):
    return s.translate(str.maketrans("ACGT", "TGCA"))[::-1]


def test_signalmachine_two_d(oracle, tmp_path):
    # --twoD: template and complement strands of a 2D R7.3 read, each with its own model; the 2D read is the reference
    # (one M run), so the complement aligns to the reverse complement (no -b: impl/fasta_handler.c:67-72)
    t_model = cases.MODEL_R73
    c_model = os.path.join(cases.GOLDEN, "models", "testModelR73_acegot_complement.model")
    npread_path = os.path.join(cases.GOLDEN, "npReads", "ZymoC_ch_1_file1.npRead")
    r = oracle.parse_npread(npread_path)
    read2d = r["twoD_read"]
    start2, L = 8, len(read2d) - 20
    ref_start = 50
    contig = "ACGT" * 12 + "AC" + read2d[start2:] + "TTGACCA" * 5
    fasta = str(tmp_path / "ref.fa")
    _write_fasta(fasta, "chr2D", contig)
    cigar = str(tmp_path / "guide.cigar")
    with open(cigar, "w") as f:
        f.write("cigar: r2d %d %d + chr2D %d %d + 1 M %d\n" % (start2, start2 + L, ref_start, ref_start + L, L))
    out = str(tmp_path / "out.tsv")
    pr = subprocess.run([BIN, "--twoD", "-T", t_model, "-C", c_model, "-q", npread_path, "-f", fasta, "-n", "chr2D", "-p", cigar,
                         "-u", out, "-L", "r2d", "-g", "100"], capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr
    assert "signalAlign - starting complement alignment" in pr.stderr and "SUCCESS" in pr.stderr
    rows = [l.rstrip("\n").split("\t") for l in open(out)]
    got = {"t": [x for x in rows if x[4] == "t"], "c": [x for x in rows if x[4] == "c"]}
    assert got["t"] and got["c"] and len(got["t"]) + len(got["c"]) == len(rows)
    assert rows.index(got["c"][0]) == len(got["t"])  # template block first, then complement (two appends)

    params = oracle.default_params()
    target_fwd = contig[ref_start:ref_start + L]
    gx, gy = oracle.guide_to_anchors(ref_start, ref_start + L, 1, start2, [(0, L)], 14)
    summary = []
    for strand, model_path in (("t", t_model), ("c", c_model)):
        om = oracle.Model.from_file(model_path)
        pre = "template" if strand == "t" else "complement"
        ev = r[pre + "_events"].copy()
        pr_ = oracle.estimate_params(om, r[pre + "_strand_event_map"], ev, r[pre + "_read"])
        em = r[pre + "_event_map"]  # 2D maps in the 2D workflow (impl/signalMachine.c:726-733)
        ax, ay = oracle.remap_anchors(gx, gy, em, start2)
        lo, hi = int(em[start2]), int(em[start2 + L - 1])
        target = target_fwd if strand == "t" else _revcomp(target_fwd)
        om.set_read_params(pr_["scale"], pr_["shift"], pr_["var"])
        pairs = oracle.align(om, target, ev[lo:hi], ax, ay, params)
        assert len(pairs) > 200
        summary.append(len(pairs))
        k, alpha, tab = om.k, om.alphabet, om.match_table()
        ref_len, ref_len_kmers = len(target), len(target) - k
        assert len(got[strand]) == len(pairs), strand
        bad = 0
        for g, p in zip(got[strand], pairs):
            x, y = int(p["x"]), int(p["y"]) + lo
            # adjustReferenceCoordinate / makeReferenceKmer for a forward-mapped read (impl/signalMachine.c:54-70)
            if strand == "t":
                x_adj, ref_kmer = x + ref_start, target[x:x + k]
            else:
                x_adj, ref_kmer = ref_len_kmers - (x + (ref_len - (ref_start + L))), _revcomp(target[x:x + k])
            kid, kmer, t_ = int(p["kmer_id"]), "", int(p["kmer_id"])
            for _ in range(k):
                kmer = alpha[t_ % len(alpha)] + kmer
                t_ //= len(alpha)
            e_mean = tab[5 * kid]
            exp = ["chr2D", str(x_adj), ref_kmer, "r2d", strand, str(y), "%f" % ev[y, 0], "%f" % ev[y, 1], "%f" % ev[y, 2],
                   target[x:x + k], "%f" % (e_mean * pr_["scale"] + pr_["shift"]), None, None,
                   "%f" % ((ev[y, 0] + pr_["var"] * e_mean - pr_["scale"] * e_mean - pr_["shift"]) / pr_["var"]),
                   "%f" % e_mean, kmer]
            for col in range(16):
                if exp[col] is not None:
                    assert g[col] == exp[col], (strand, col, g, exp)
            d = abs(float(g[12]) - int(p["prob_e7"]) / 1e7)
            assert d <= 1e-5
            bad += d > 1.5e-6  # the column is printed with six decimals
        assert bad <= len(pairs) // 100
    head = pr.stdout.strip("\n").split("\t")
    assert head[1].startswith("%d(" % summary[0]) and head[2].startswith("%d(" % summary[1])


def test_signalmachine_minus_strand_and_assignments(oracle, tmp_path):
    # a read mapped to the reverse strand (cigar with '-', rstart > rend: utils/bwaWrapper.py:213-214) with an explicit
    # backward reference (-b), full TSV (-s 0) and the assignments rendering (-s 2)
    model = cases.MODEL_5MER
    npread_path = os.path.join(cases.GOLDEN, "npReads", "c2925_ecoli_ch34_read1023.npRead")
    r = oracle.parse_npread(npread_path)
    read = r["template_read"]
    start2, L = 6, len(read) - 14
    part = read[start2:start2 + L]
    pre, post = "GATTACA" * 9, "CCGGTTAA" * 6
    contig = pre + _revcomp(part) + post          # the read matches the reverse strand of the contig
    fasta, bfasta = str(tmp_path / "fwd.fa"), str(tmp_path / "bwd.fa")
    _write_fasta(fasta, "chrM", contig)
    _write_fasta(bfasta, "chrM", contig.translate(str.maketrans("ACGT", "TGCA")))  # the complement, same coordinates
    end1, start1 = len(pre), len(pre) + L
    cigar = str(tmp_path / "guide.cigar")
    with open(cigar, "w") as f:
        f.write("cigar: rm %d %d + chrM %d %d - 1 M %d\n" % (start2, start2 + L, start1, end1, L))
    om = oracle.Model.from_file(model)
    ev = r["template_events"].copy()
    pr_ = oracle.estimate_params(om, r["template_strand_event_map"], ev, read)
    em = r["template_strand_event_map"]
    gx, gy = oracle.guide_to_anchors(start1, end1, 0, start2, [(0, L)], 14)
    ax, ay = oracle.remap_anchors(gx, gy, em, start2)
    lo, hi = int(em[start2]), int(em[start2 + L - 1])
    target = part                                   # reverse of the complement window == the read's own bases
    om.set_read_params(pr_["scale"], pr_["shift"], pr_["var"])
    pairs = oracle.align(om, target, ev[lo:hi], ax, ay, oracle.default_params())
    assert len(pairs) > 300
    k, alpha, tab = om.k, om.alphabet, om.match_table()

    full, assign = str(tmp_path / "full.tsv"), str(tmp_path / "assign.tsv")
    common = [BIN, "-T", model, "-q", npread_path, "-f", fasta, "-b", bfasta, "-n", "chrM", "-p", cigar, "-L", "rm", "-g", "100"]
    for fmt, out in (("0", full), ("2", assign)):
        pr = subprocess.run(common + ["-s", fmt, "-u", out], capture_output=True, text=True, timeout=300)
        assert pr.returncode == 0, pr.stderr
    rows = [l.rstrip("\n").split("\t") for l in open(full)]
    arows = [l.rstrip("\n").split("\t") for l in open(assign)]
    fc.check_assignment_rows(open(assign).read(), len(arows[0][0]), "ACGT")   # as the reference's golden assignments file
    assert len(rows) == len(pairs) == len(arows)
    ref_len, ref_len_kmers = len(target), len(target) - k
    for g, a, p in zip(rows, arows, pairs):
        x, y = int(p["x"]), int(p["y"]) + lo
        kid, kmer, t_ = int(p["kmer_id"]), "", int(p["kmer_id"])
        for _ in range(k):
            kmer = alpha[t_ % len(alpha)] + kmer
            t_ //= len(alpha)
        e_mean = tab[5 * kid]
        desc = (ev[y, 0] + pr_["var"] * e_mean - pr_["scale"] * e_mean - pr_["shift"]) / pr_["var"]
        # template strand of a reverse-mapped read: mirrored coordinate, reverse-complemented reference k-mer
        x_adj = ref_len_kmers - (x + (ref_len - start1))
        assert g[:6] == ["chrM", str(x_adj), _revcomp(target[x:x + k]), "rm", "t", str(y)], (g, x, y)
        assert g[9] == target[x:x + k] and g[13] == "%f" % desc and g[15] == kmer
        assert abs(float(g[12]) - int(p["prob_e7"]) / 1e7) <= 1e-5
        assert a[0] == kmer and a[1] == "t" and a[2] == "%f" % desc
        assert abs(float(a[3]) - int(p["prob_e7"]) / 1e7) <= 1e-5


def test_signalmachine_hdp_model(oracle, tmp_path):
    # -v <.nhdp> switches to the HMM-HDP state machine (impl/signalMachine.c:667-690): emissions from the HDP, the
    # model means replaced by the HDP's expected values before alignment (:861-863) -- which the E_mean column shows
    model, nhdp = cases.MODEL_R73, cases.NHDP
    npread_path = os.path.join(cases.GOLDEN, "npReads", "ZymoC_ch_1_file1.npRead")
    r = oracle.parse_npread(npread_path)
    read = r["template_read"]
    start2, L = 4, len(read) - 12
    fasta = str(tmp_path / "ref.fa")
    _write_fasta(fasta, "chrH", "TTTT" + read[start2:] + "ACACAC")
    cigar = str(tmp_path / "guide.cigar")
    with open(cigar, "w") as f:
        f.write("cigar: rh %d %d + chrH 4 %d + 1 M %d\n" % (start2, start2 + L, 4 + L, L))
    out = str(tmp_path / "out.tsv")
    pr = subprocess.run([BIN, "-T", model, "-v", nhdp, "-q", npread_path, "-f", fasta, "-n", "chrH", "-p", cigar, "-u", out,
                         "-L", "rh", "-g", "100", "-D", "0.05"], capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr
    assert "Using threeStateHdp stateMachine since you pass in an HDP file" in pr.stderr
    # the HDP path at the caller's default threshold is where the pairs outweigh the kernels (17.8 pairs per event on the bundled
    # model): full and assignments output on 8-byte records (the default) and on 16-byte records give the same bytes
    for fmt in ("0", "2"):
        a8, a16 = str(tmp_path / ("p8_%s.tsv" % fmt)), str(tmp_path / ("p16_%s.tsv" % fmt))
        for path, env in ((a8, dict(os.environ)), (a16, dict(os.environ, SA_CLI_PAIRS16="1"))):
            q = subprocess.run([BIN, "-T", model, "-v", nhdp, "-q", npread_path, "-f", fasta, "-n", "chrH", "-p", cigar, "-u", path,
                                "-L", "rh", "-g", "100", "-D", "0.01", "-s", fmt], capture_output=True, text=True, timeout=300, env=env)
            assert q.returncode == 0, q.stderr
        assert open(a8, "rb").read() == open(a16, "rb").read() and len(open(a8).readlines()) > 1000
    om = oracle.Model.from_file(model)
    ev = r["template_events"].copy()
    pr_ = oracle.estimate_params(om, r["template_strand_event_map"], ev, read)  # before the HDP means are installed
    om.load_hdp(nhdp)
    om.set_to_hdp_expected_values()
    em = r["template_strand_event_map"]
    gx, gy = oracle.guide_to_anchors(4, 4 + L, 1, start2, [(0, L)], 14)
    ax, ay = oracle.remap_anchors(gx, gy, em, start2)
    lo, hi = int(em[start2]), int(em[start2 + L - 1])
    om.set_read_params(pr_["scale"], pr_["shift"], pr_["var"])
    pairs = oracle.align(om, read[start2:start2 + L], ev[lo:hi], ax, ay, oracle.default_params(threshold=0.05))
    rows = [l.rstrip("\n").split("\t") for l in open(out)]
    assert len(pairs) > 50
    exp = {(int(p["x"]) + 4, int(p["y"]) + lo): p for p in pairs}
    got = {(int(g[1]), int(g[5])): g for g in rows}
    tab = om.match_table()
    lonely = 0
    for key in set(exp) | set(got):
        if key in exp and key in got:
            p, g = exp[key], got[key]
            assert abs(float(g[12]) - int(p["prob_e7"]) / 1e7) <= 1e-5 + 5e-7
            assert g[14] == "%f" % tab[5 * int(p["kmer_id"])]  # HDP expected mean, not the .model file's
        else:
            pv = int(exp[key]["prob_e7"]) / 1e7 if key in exp else float(got[key][12])
            assert abs(pv - 0.05) <= 2e-5, key  # only pairs sitting on the threshold may differ
            lonely += 1
    assert lonely <= 2


@pytest.mark.parametrize("minus", [False, True])
def test_signalmachine_mea_output(oracle, tmp_path, minus):
    """--mea writes <posteriors>.mea: the rows of the full TSV on the maximum-expected-accuracy path.  Checked the way
    the reference would produce it (mea_alignment_from_signal_align, src/signalalign/mea_algorithm.py:323-341): read
    the TSV the binary just wrote, take its reference_index / event_index / posterior columns, run
    get_mea_params_from_events + maximum_expected_accuracy_alignment (CPU restatement) and keep those rows -- for a
    forward and for a reverse-strand read (where the table's reference indices fall along the read and the reference
    flips them)."""
    model = cases.MODEL_5MER
    npread_path = os.path.join(cases.GOLDEN, "npReads", "c2925_ecoli_ch34_read1023.npRead")
    r = oracle.parse_npread(npread_path)
    read = r["template_read"]
    start2, L = 6, len(read) - 14
    part = read[start2:start2 + L]
    pre, post = "GATTACA" * 9, "CCGGTTAA" * 6
    fasta = str(tmp_path / "fwd.fa")
    cigar = str(tmp_path / "guide.cigar")
    argv = [BIN, "-T", model, "-q", npread_path, "-n", "chrM", "-p", cigar, "-L", "rm", "-g", "100", "-s", "0", "--mea"]
    if minus:
        contig = pre + _revcomp(part) + post
        bfasta = str(tmp_path / "bwd.fa")
        _write_fasta(bfasta, "chrM", contig.translate(str.maketrans("ACGT", "TGCA")))
        with open(cigar, "w") as f:
            f.write("cigar: rm %d %d + chrM %d %d - 1 M %d\n" % (start2, start2 + L, len(pre) + L, len(pre), L))
        argv += ["-b", bfasta]
    else:
        contig = pre + part + post
        with open(cigar, "w") as f:
            f.write("cigar: rm %d %d + chrM %d %d + 1 M %d\n" % (start2, start2 + L, len(pre), len(pre) + L, L))
    _write_fasta(fasta, "chrM", contig)
    out = str(tmp_path / "post.tsv")
    pr = subprocess.run(argv + ["-f", fasta, "-u", out], capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr
    rows = [l.rstrip("\n").split("\t") for l in open(out)]
    mea_rows = [l.rstrip("\n").split("\t") for l in open(out + ".mea")]
    ref_index = np.array([int(g[1]) for g in rows])
    event_index = np.array([int(g[5]) for g in rows])
    posterior = np.array([float(g[12]) for g in rows])
    if minus:
        assert ref_index[0] > ref_index[-1]                 # the table really runs backwards along the reference
    ev, rf, po, sh = oracle.mea_params(ref_index, event_index, posterior)
    st, path, best = oracle.mea(ev, rf, po, sh)
    assert st == 0 and len(path) > 300
    # back from matrix indices to the table's own columns
    e0 = event_index.min()
    r_of = (lambda c: ref_index.max() - c) if minus else (lambda c: ref_index.min() + c)
    want = {(r_of(int(c)), int(e) + e0) for c, e in path}
    got = [(int(g[1]), int(g[5])) for g in mea_rows]
    assert len(got) == len(set(got)) == len(want) and set(got) == want
    # and they are rows of the posteriors file, in its order
    index_of = {tuple(g): i for i, g in enumerate(rows)}
    where = [index_of[tuple(g)] for g in mea_rows]
    assert where == sorted(where)
