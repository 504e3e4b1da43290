"""CPU-only checks of the product's host side (planning, anchors, parameter estimation, loaders, ABI)
against the oracle.  No GPU compute is called here."""
import os
import re

import numpy as np
import pytest

import signalalign_amd as sa
from signalalign_amd import synth
from signalalign_amd._capi import EXPORTS

import sa_cases as cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "signalalign_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sa_[a-z0-9_]+)\s*\(", hdr))
    # (the pack / unpack helpers of the 16-byte result record are `static inline` in the header itself, not exports)
    declared -= set(re.findall(r"SA_PAIR16_FN\s+\w+\s+(sa_[a-z0-9_]+)\s*\(", hdr))
    L = sa.lib()
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert declared == set(EXPORTS), declared ^ set(EXPORTS)


def test_no_device_fails_loudly():
    # this suite runs without a GPU: the compute entry points must refuse, never fall back
    if sa.device_count() > 0:
        pytest.skip("a GPU is present")
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_5MER)
    m = sa.Model.create(alpha, k, t10, tab)
    job = dict(ref="ACGTACGTACGT", events=np.array([60.0, 61.0]), ax=[], ay=[], scale=1.0, shift=0.0, var=1.0)
    with pytest.raises(sa.SaError) as ei:
        sa.Batch(m, sa.default_params(), [job])
    assert ei.value.code == -3


def test_model_loader_matches_text(oracle):
    m = sa.Model.load(cases.MODEL_6MER)
    d = oracle.parse_model_file(cases.MODEL_6MER)
    assert m.alphabet() == ("ACGT", 6)
    assert np.array_equal(m.table5(), d["table5"])
    assert m.kmer_id("AAAAAC") == 1 and m.kmer_id("AAAANC") == -1
    with pytest.raises(sa.SaError):
        sa.Model.load(os.path.join(cases.GOLDEN, "npReads", "ZymoRef.txt"))


def test_hdp_loader_and_expected_values(oracle):
    m = sa.Model.load(cases.MODEL_R73, cases.NHDP)
    om = oracle.Model.from_file(cases.MODEL_R73)
    om.load_hdp(cases.NHDP)
    m.set_to_hdp_expected_values()
    om.set_to_hdp_expected_values()
    assert np.array_equal(m.table5(), om.match_table())


@pytest.mark.parametrize("n_events,thin", [(300, 0.0), (2500, 0.0), (2500, 0.3)])
def test_plan_geometry_matches_oracle(oracle, n_events, thin):
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    m = sa.Model.create(alpha, k, t10, tab)
    om = oracle.Model(alpha, k, t10, tab)
    p = sa.default_params()
    op = cases.oracle_params(oracle, p)
    for idx in range(3):
        job = synth.make_read(idx, n_events, alpha, k, tab, thin_anchors=thin)
        info, regions, rows, segs = sa.plan_describe(m, p, job)
        lX, lY = len(job["ref"]) - (k - 1), len(job["events"])
        exp_regions = oracle.split_points(job["ax"], job["ay"], lX, lY, p.split_matrix_bigger_than_this, 1, 1)
        assert np.array_equal(regions, exp_regions)
        # band rows of every region
        off = 0
        for r, (x1, y1, x2, y2) in enumerate(regions):
            sel = (job["ax"] + job["ay"] >= x1 + y1) & (job["ax"] + job["ay"] < x2 + y2)
            L, R = oracle.band(job["ax"][sel] - x1, job["ay"][sel] - y1, x2 - x1, y2 - y1, p.diagonal_expansion)
            n = len(L)
            assert np.array_equal(rows[off:off + n, 0], np.full(n, r))
            assert np.array_equal(rows[off:off + n, 1], L)
            assert np.array_equal(rows[off:off + n, 2], R)
            off += n
        # the work the oracle actually performs equals what the plan announces
        om.set_read_params(job["scale"], job["shift"], job["var"])
        _, st = oracle.align(om, job["ref"], job["events"], job["ax"], job["ay"], op, want_stats=True)
        assert info.cells_forward == st.cells_forward
        assert info.cells_backward == st.cells_backward
        assert info.n_segments == st.n_tracebacks


def test_split_regions_large_gap(oracle):
    # anchors with a > 3000x3000 hole: the plan must cut exactly where getSplitPoints cuts
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    m = sa.Model.create(alpha, k, t10, tab)
    p = sa.default_params()
    job = synth.make_read(7, 14000, alpha, k, tab)
    hole = (job["ax"] > 1500) & (job["ax"] < 6500)
    job["ax"], job["ay"] = job["ax"][~hole], job["ay"][~hole]
    info, regions, rows, segs = sa.plan_describe(m, p, job)
    lX, lY = len(job["ref"]) - (k - 1), len(job["events"])
    exp = oracle.split_points(job["ax"], job["ay"], lX, lY, p.split_matrix_bigger_than_this, 1, 1)
    assert len(exp) == 2 and np.array_equal(regions, exp)


def test_split_regions_follow_the_ragged_end_flags():
    # tests/signalPairwiseAlignerTest.c:363-432 test_getSplitPoints, the reference's literal cases, through the PRODUCT's host
    # planner (sa_job_t.ends = getAlignedPairsUsingAnchors' two ragged-end booleans, inverted): no oracle involved
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    m = sa.Model.create(alpha, k, t10, tab)
    p = sa.default_params(split=2000 * 2000)
    rng = np.random.default_rng(3)

    def regions(lX, lY, ax, ay, ragged):
        job = dict(ref="".join(rng.choice(list("ACGT"), lX + k - 1)), events=np.full(lY, 80.0), ax=ax, ay=ay, ragged=ragged)
        return sa.plan_describe(m, p, job)[1].tolist()

    assert regions(3000, 1000, [], [], (0, 0)) == [[0, 0, 3000, 1000]]
    lX, lY = 20000, 25000
    assert regions(lX, lY, [], [], (1, 1)) == []
    assert regions(lX, lY, [], [], (1, 0)) == [[18000, 23000, lX, lY]]
    assert regions(lX, lY, [], [], (0, 1)) == [[0, 0, 2000, 2000]]
    assert regions(lX, lY, [], [], (0, 0)) == [[0, 0, 2000, 2000], [18000, 23000, lX, lY]]
    ax = [2000, 4002, 5000, 8000, 9000, 10000, 15000, 16000]
    ay = [2000, 4001, 5000, 6000, 9000, 14000, 15000, 16000]
    assert regions(lX, lY, ax, ay, (0, 0)) == [[0, 0, 3001, 3001], [3002, 3001, 9500, 11001], [9501, 12000, 12001, 14500],
                                               [13000, 14501, 18000, 18001], [18001, 23000, 20000, 25000]]
    # the default (a zeroed `ends`) is signalMachine's call, ragged on both sides: the rectangle behind the last cut goes (the gap
    # behind the last anchor is cut), the first one stays (the gap in front of the FIRST anchor, 2000 x 2000, is not cut)
    assert regions(lX, lY, ax, ay, (1, 1)) == [[0, 0, 3001, 3001], [3002, 3001, 9500, 11001], [9501, 12000, 12001, 14500],
                                               [13000, 14501, 18000, 18001]]
    assert regions(lX, lY, [4000] + ax[1:], [4000] + ay[1:], (1, 0))[0] == [2000, 2000, 9500, 11001]   # (the first anchor at 4000, 4000: its gap is cut and the ragged left end drops the rectangle in front of it)
    job = dict(ref="A" * 30, events=np.full(9, 80.0), ax=[], ay=[])
    assert sa.plan_describe(m, p, job)[0].n_regions == sa.plan_describe(m, p, dict(job, ragged=(1, 1)))[0].n_regions == 1


def test_anchor_helpers_match_oracle(oracle):
    rng = np.random.default_rng(3)
    for trial in range(20):
        ops, ref_len, read_len = [], 0, 0
        for _ in range(rng.integers(1, 12)):
            t = int(rng.integers(0, 3))
            ln = int(rng.integers(1, 90))
            ops.append((t, ln))
            if t != 2:
                ref_len += ln
            if t != 1:
                read_len += ln
        strand = int(rng.integers(0, 2))
        start1 = int(rng.integers(0, 1000))
        s1, e1 = (start1, start1 + ref_len) if strand else (start1 + ref_len, start1)
        start2 = int(rng.integers(0, 50))
        trim = int(rng.integers(0, 20))
        a = sa.guide_to_anchors(s1, e1, strand, start2, ops, trim)
        b = oracle.guide_to_anchors(s1, e1, strand, start2, ops, trim)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        if len(a[0]) == 0:
            continue
        emap = np.cumsum(rng.integers(0, 4, size=start2 + read_len + 1))
        c = sa.remap_anchors(a[0], a[1], emap, start2)
        d = oracle.remap_anchors(a[0], a[1], emap, start2)
        assert np.array_equal(c[0], d[0]) and np.array_equal(c[1], d[1])
        # filterToRemoveOverlap (impl/pairwiseAligner.c:1755-1796): what is left rises strictly in both coordinates
        assert np.all(np.diff(c[0]) > 0) and np.all(np.diff(c[1]) > 0)


@pytest.mark.parametrize("name,model", [("r9p4_oneD.npRead", cases.MODEL_6MER),
                                        ("c2925_ecoli_ch34_read1023.npRead", cases.MODEL_5MER)])
def test_estimate_params_matches_oracle(oracle, name, model):
    r = oracle.parse_npread(os.path.join(cases.GOLDEN, "npReads", name))
    om = oracle.Model.from_file(model)
    ev_o = r["template_events"].copy()
    exp = oracle.estimate_params(om, r["template_strand_event_map"], ev_o, r["template_read"])
    m = sa.Model.load(model)
    tab = m.table5().copy()
    ev_p = r["template_events"].copy()
    got = sa.estimate_params(m, tab, r["template_strand_event_map"], ev_p, r["template_read"])
    assert got == exp
    assert np.array_equal(ev_p, ev_o)
    assert np.array_equal(tab, om.match_table())


def test_threaded_planner_is_deterministic():
    # the planner cuts the job list into ranges, plans them in threads and concatenates: every uploaded array must be
    # bit-identical to what one thread produces, whatever the thread count and wherever the cuts fall
    pm = sa.Model.load(cases.MODEL_6MER)
    p = sa.default_params()
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 23, 600) + cases.synthetic_jobs(cases.MODEL_6MER, 3, 2600, 100)
    jobs.insert(5, dict(ref="ACGTACGT", events=np.zeros(0), ax=[], ay=[]))  # a job without events in the middle
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    big = synth.make_read(7, 9000, alpha, k, tab)
    hole = (big["ax"] > 1500) & (big["ax"] < 5500)
    big["ax"], big["ay"] = big["ax"][~hole], big["ay"][~hole]
    jobs.insert(11, big)  # splits into two regions
    ref_info, ref_digest = sa.plan_digest(pm, p, jobs, threads=1)
    assert ref_info.n_regions == len(jobs)  # 28 with events + 1 extra region of the split read - 1 empty job
    for t in (2, 3, 5, 7):
        info, digest = sa.plan_digest(pm, p, jobs, threads=t)
        assert digest == ref_digest, t
        assert (info.n_regions, info.n_segments, info.n_checkpoints) == (ref_info.n_regions, ref_info.n_segments,
                                                                          ref_info.n_checkpoints)
        assert info.cells_forward == ref_info.cells_forward and info.cells_backward == ref_info.cells_backward
    # the memory-resident variant lays its rows out differently: a different digest, equally stable
    g1 = sa.plan_digest(pm, p, jobs, flags=sa.FLAG_FORCE_GENERIC, threads=1)[1]
    g4 = sa.plan_digest(pm, p, jobs, flags=sa.FLAG_FORCE_GENERIC, threads=4)[1]
    assert g1 == g4
    # round 4: an HDP model with ambiguity letters plans ring-kernel regions with per-path records in every thread's range
    hd = sa.Model.load(cases.MODEL_R73, cases.NHDP)
    hj = [dict(j, ref=j["ref"].replace("CG", "LG")) for j in cases.synthetic_jobs(cases.MODEL_R73, 9, 500, 40)]
    h1 = sa.plan_digest(hd, p, hj, threads=1)
    h3 = sa.plan_digest(hd, p, hj, threads=3)
    assert h1[1] == h3[1] and h1[0].n_ring_regions == len(hj)


def test_planner_error_codes_and_empty_batch():
    # the reference asserts / aborts on these (impl/pairwiseAligner.c:1460-1464, :195-246); the library returns codes
    pm = sa.Model.load(cases.MODEL_6MER)
    p = sa.default_params()
    info, digest = sa.plan_digest(pm, p, [], threads=1)           # nothing to align is not an error
    assert (info.n_regions, info.n_segments) == (0, 0)
    job = cases.synthetic_jobs(cases.MODEL_6MER, 1, 300)[0]
    bad = dict(job)
    bad["ax"], bad["ay"] = job["ax"][::-1].copy(), job["ay"][::-1].copy()   # anchors must increase in both coordinates
    with pytest.raises(sa.SaError) as ei:
        sa.plan_digest(pm, p, [bad], threads=1)
    assert ei.value.code == -5                                    # SA_EBAND
    out_of_range = dict(job)
    out_of_range["ax"] = job["ax"] + 10 ** 6
    with pytest.raises(sa.SaError) as ei:
        sa.plan_digest(pm, p, [out_of_range], threads=1)
    assert ei.value.code == -5
    for kw in (dict(trace_back=1000), dict(expansion=51), dict(threshold=1.5)):
        q = sa.default_params(**kw)
        if "expansion" in kw:
            q.diagonal_expansion = 51                             # default_params() would round it up, as signalMachine does
        with pytest.raises(sa.SaError) as ei:
            sa.plan_digest(pm, q, [job], threads=1)
        assert ei.value.code == -1, kw                            # SA_EINVAL
    # the first failing job in job order decides, whatever the thread count
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 12, 200)
    jobs[7] = bad
    for t in (1, 3):
        with pytest.raises(sa.SaError) as ei:
            sa.plan_digest(pm, p, jobs, threads=t)
        assert ei.value.code == -5


def test_cells_the_result_record_cannot_name_are_refused():
    # results travel as 16-byte records with 16 bits of path index (sa_pair16_t): an ambiguity letter with eight options at
    # six adjacent positions of a 6-mer window gives a cell 8^6 = 262144 paths -- refused by the planner (SA_EUNSUPPORTED)
    # instead of truncated; the same letter at five positions (32768 paths) is planned
    pm = sa.Model.load(cases.MODEL_6MER)
    p = sa.default_params()
    amb = sa.default_ambig({"Z": "ACGTACGT"})
    job = cases.synthetic_jobs(cases.MODEL_6MER, 1, 120)[0]
    ref = job["ref"]
    too_many = dict(job)
    too_many["ref"] = ref[:30] + "ZZZZZZ" + ref[36:]
    with pytest.raises(sa.SaError) as ei:
        sa.plan_digest(pm, p, [too_many], ambig=amb, threads=1)
    assert ei.value.code == -8                                    # SA_EUNSUPPORTED
    fits = dict(job)
    fits["ref"] = ref[:30] + "ZZZZZ" + ref[35:]
    info, _ = sa.plan_digest(pm, p, [fits], ambig=amb, threads=1)
    assert info.n_regions == 1


def test_scalings_by_method_of_moments(oracle):
    # estimate_scalings_using_mom (impl/eventAligner.c:784-843): host-only entry point, bit-identical to the restatement
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    om = oracle.Model(alpha, k, t10, tab)
    pm = sa.Model.load(cases.MODEL_6MER)
    r = synth.make_read(31, 1200, alpha, k, tab)
    ev = np.ascontiguousarray(np.asarray(r["events4"])[:, 0])
    sh, sc = sa.scalings_mom(pm, r["ref"], ev)
    osh, osc = oracle.scalings_mom(om, ev, oracle.kmer_ids_of(om, r["ref"]))
    assert (sh, sc) == (osh, osc)
    assert abs(sc - 1.0) < 0.15 and abs(sh) < 15.0
    with pytest.raises(sa.SaError) as ei:
        sa.scalings_mom(pm, "ACGTNACGTACGT", ev)        # a letter outside the model's alphabet
    assert ei.value.code == -4


def test_plan_geometry_random_anchor_sets(oracle):
    """Many small matrices with arbitrary (sorted, strictly increasing) anchor sets -- dense, sparse, hugging a border,
    none at all -- and several band expansions / split limits: regions, every band row and the announced work equal the
    oracle's.  Complements the synthetic-read cases above, whose anchors always follow the read's own event map."""
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_5MER)
    m = sa.Model.create(alpha, k, t10, tab)
    om = oracle.Model(alpha, k, t10, tab)
    rng = np.random.default_rng(2024)
    for it in range(60):
        n_events = int(rng.integers(1, 400))
        job = synth.make_read(int(rng.integers(0, 10 ** 6)), n_events, alpha, k, tab)
        lX, lY = len(job["ref"]) - (k - 1), len(job["events"])
        n_anchor = int(rng.integers(0, min(lX, lY) + 1))
        style = it % 4
        if style == 0 or n_anchor == 0 or lX < 2 or lY < 2:
            ax = ay = np.zeros(0, dtype=np.int64)
        else:
            ax = np.sort(rng.choice(lX, size=min(n_anchor, lX), replace=False))
            ay = np.sort(rng.choice(lY, size=min(n_anchor, lY), replace=False))
            n = min(len(ax), len(ay))
            ax, ay = ax[:n].astype(np.int64), ay[:n].astype(np.int64)
            if style == 2:
                ay = np.minimum(ay, np.arange(n))          # hugging the lower border
                ay = np.maximum.accumulate(ay)
                keep = np.concatenate([[True], np.diff(ay) > 0])
                ax, ay = ax[keep], ay[keep]
        job["ax"], job["ay"] = ax, ay
        p = sa.default_params(expansion=int(rng.choice([0, 2, 10, 50])), trace_back=int(rng.choice([5, 30, 100])),
                              min_diags=int(rng.choice([120, 1000])), split=int(rng.choice([40 * 40, 3000 * 3000])))
        if p.trace_back_diagonals + 1 >= p.min_diags_between_trace_back:
            continue
        op = cases.oracle_params(oracle, p)
        info, regions, rows, segs = sa.plan_describe(m, p, job)
        exp_regions = oracle.split_points(ax, ay, lX, lY, p.split_matrix_bigger_than_this, 1, 1)
        assert np.array_equal(regions, exp_regions), it
        off = 0
        for r, (x1, y1, x2, y2) in enumerate(regions):
            sel = (ax + ay >= x1 + y1) & (ax + ay < x2 + y2)
            L, R = oracle.band(ax[sel] - x1, ay[sel] - y1, x2 - x1, y2 - y1, p.diagonal_expansion)
            n = len(L)
            assert np.array_equal(rows[off:off + n, 1], L) and np.array_equal(rows[off:off + n, 2], R), (it, r)
            off += n
        om.set_read_params(job["scale"], job["shift"], job["var"])
        _, st = oracle.align(om, job["ref"], job["events"], ax, ay, op, want_stats=True)
        assert info.cells_forward == st.cells_forward and info.cells_backward == st.cells_backward, it
        assert info.n_segments == st.n_tracebacks, it


def test_fasta_subsequence_reference_known_answer(tmp_path):
    """tests/fastaHandlerTests.c:15-32 (test_fastaHandler_getSubSequence): bases [0, 10) of record ZYMO of the reference's
    own fixture are AGAATTGGTT -- with an index file next to the FASTA and, as htslib would build one, without."""
    import ctypes as C
    import shutil
    L = sa.lib()

    def fetch(path, name, start, end, strand=1):
        out = C.c_void_p()
        rc = L.sa_fasta_subsequence(path.encode(), name.encode(), start, end, strand, C.byref(out))
        if rc:
            return rc
        s = C.string_at(out).decode()
        L.sa_free(out)
        return s

    src = os.path.join(cases.GOLDEN, "sequences", "pUC19_SspI_Zymo.fa")
    fa = str(tmp_path / "ref.fa")
    shutil.copy(src, fa)
    text = open(fa).read()
    records, name, seq = {}, None, []
    for line in text.splitlines():
        if line.startswith(">"):
            if name is not None:
                records[name] = "".join(seq)
            name, seq = line[1:].split()[0], []
        else:
            seq.append(line.strip())
    records[name] = "".join(seq)
    assert fetch(fa, "ZYMO", 0, 10) == "AGAATTGGTT"                      # the reference's assertion, no .fai present
    for rec, s in records.items():
        for a, b in ((0, 10), (5, 77), (len(s) - 9, len(s)), (61, 200), (0, len(s))):
            assert fetch(fa, rec, a, b) == s[a:b], (rec, a, b)
    assert fetch(fa, "nope", 0, 10) != 0 and not isinstance(fetch(fa, "nope", 0, 10), str)
    # the same through a faidx index (name, length, offset, bases per line, bytes per line)
    with open(fa + ".fai", "w") as f:
        off = 0
        lines = text.splitlines(True)
        i = 0
        while i < len(lines):
            if lines[i].startswith(">"):
                nm = lines[i][1:].split()[0]
                off += len(lines[i])
                first = lines[i + 1]
                f.write("%s\t%d\t%d\t%d\t%d\n" % (nm, len(records[nm]), off, len(first.rstrip("\n")), len(first)))
                i += 1
                while i < len(lines) and not lines[i].startswith(">"):
                    off += len(lines[i])
                    i += 1
            else:
                i += 1
    assert fetch(fa, "ZYMO", 0, 10) == "AGAATTGGTT"
    for rec, s in records.items():
        assert fetch(fa, rec, 61, 200) == s[61:200]


def test_cigar_loader_on_the_reference_guide_alignment(oracle):
    """The exonerate cigar line the reference's own test expects from its bwa wrapper for a reverse-strand E. coli read
    (src/signalalign/tests/test_bwaWrapper.py:42-47, kept as tests/golden/cigars/ecoli_minus_strand.cigar): the
    product's loader reads it as sonLib's cigarRead would (coordinates, strands, 1261 operations) and the anchors
    derived from it equal the oracle's."""
    import ctypes as C

    class Cigar(C.Structure):
        _fields_ = [("contig1", C.c_char_p), ("contig2", C.c_char_p), ("start1", C.c_int64), ("end1", C.c_int64),
                    ("start2", C.c_int64), ("end2", C.c_int64), ("strand1", C.c_int), ("strand2", C.c_int),
                    ("score", C.c_double), ("n_ops", C.c_int64), ("op_type", C.POINTER(C.c_int32)),
                    ("op_len", C.POINTER(C.c_int64))]
    L = sa.lib()
    path = os.path.join(cases.GOLDEN, "cigars", "ecoli_minus_strand.cigar")
    pc = C.POINTER(Cigar)()
    L.sa_cigar_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(Cigar))]
    L.sa_cigar_free.argtypes = [C.POINTER(Cigar)]
    assert L.sa_cigar_load(path.encode(), C.byref(pc)) == 0
    c = pc.contents
    toks = open(path).read().split()
    assert c.contig2 == toks[1].encode() and c.contig1 == b"gi_ecoli"
    assert (c.start2, c.end2, c.strand2) == (1, 11458, 1) and (c.start1, c.end1, c.strand1) == (1845113, 1832930, 0)
    assert c.score == 1.0 and c.n_ops == (len(toks) - 10) // 2 == 1261
    ops = [(int(c.op_type[i]), int(c.op_len[i])) for i in range(c.n_ops)]
    letters = {"M": 0, "D": 1, "I": 2}
    assert ops == [(letters[toks[10 + 2 * i]], int(toks[11 + 2 * i])) for i in range(c.n_ops)]
    m = sum(n for t, n in ops if t == 0)
    d = sum(n for t, n in ops if t == 1)
    ins = sum(n for t, n in ops if t == 2)
    assert m + ins == c.end2 - c.start2 and m + d == c.start1 - c.end1      # both spans add up: 11457 read, 12183 reference bases
    for trim in (0, 14):
        a = sa.guide_to_anchors(c.start1, c.end1, c.strand1, c.start2, ops, trim)
        b = oracle.guide_to_anchors(c.start1, c.end1, c.strand1, c.start2, ops, trim)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        # untrimmed, every matched base is an anchor except those within 6 of the reference end (impl/pairwiseAligner.c
        # :1640-1652); the default trim of 14 at both ends of every match run leaves a sixth of them (a real alignment has an
        # indel every ten to fifty bases)
        assert len(a[0]) == (m - 5 if trim == 0 else 1801)
    L.sa_cigar_free(pc)


def test_loaders_reject_malformed_files(tmp_path):
    """Malformed .model / .npRead / .nhdp files are refused with an error code -- no read past a short token list, no
    write in front of a buffer for a negative declared length, no k-mer walk over a sequence shorter than declared.
    (probes/host_asan.sh runs this file under AddressSanitizer + UBSan.)"""
    import ctypes as C
    L = sa.lib()
    good = open(cases.MODEL_5MER).read().split("\n")
    # a 2-state header with the 5 tokens that would pass a "n_states^2 + 1" check: must not index ten transitions
    bad_model = tmp_path / "two_state.model"
    bad_model.write_text("2\t4\tACGT\t5\n0.1 0.2 0.3 0.4 0.5\n" + good[2] + "\n")
    h = C.c_void_p()
    assert L.sa_model_load(C.byref(h), str(bad_model).encode(), None) != 0 and not h.value
    short_table = tmp_path / "short_table.model"
    short_table.write_text(good[0] + "\n" + good[1] + "\n" + " ".join(good[2].split()[:-5]) + "\n")
    assert L.sa_model_load(C.byref(h), str(short_table).encode(), None) != 0 and not h.value
    L.sa_npread_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.sa_npread_load.restype = C.c_int
    L.sa_npread_free.argtypes = [C.c_void_p]
    src = open(os.path.join(cases.GOLDEN, "npReads", "c2925_ecoli_ch34_read1023.npRead")).read().split("\n")
    r = C.c_void_p()
    ok = tmp_path / "ok.npRead"
    ok.write_text("\n".join(src))
    assert L.sa_npread_load(str(ok).encode(), C.byref(r)) == 0 and r.value
    L.sa_npread_free(r)

    def variant(name, edit):
        lines = list(src)
        edit(lines)
        p = tmp_path / name
        p.write_text("\n".join(lines))
        out = C.c_void_p()
        rc = L.sa_npread_load(str(p).encode(), C.byref(out))
        assert rc != 0 and not out.value, name

    def header(lines, idx, value):
        t = lines[0].split()
        t[idx] = str(value)
        lines[0] = " ".join(t)
    variant("neg_template_len.npRead", lambda l: header(l, 3, -1))
    variant("neg_events.npRead", lambda l: header(l, 1, -4))
    variant("short_template_read.npRead", lambda l: l.__setitem__(2, l[2].strip()[:-7]))
    variant("short_event_map.npRead", lambda l: l.__setitem__(3, " ".join(l[3].split()[:-1])))
    variant("truncated.npRead", lambda l: l.__delitem__(slice(6, None)))
    # .nhdp: a zero grid length, and a file that ends inside the parent table (the dp-id list is already allocated)
    nh = open(cases.NHDP).read().split("\n")
    grid_line = 10                              # "grid_start grid_stop grid_length" (serialize_hdp, impl/hdp.c:2919-3050)
    assert nh[grid_line].split() == ["0", "100", "100"]
    for name, edit in (("zero_grid.nhdp", lambda l: l.__setitem__(grid_line, "0.0 100.0 0")),
                       ("cut.nhdp", lambda l: l.__delitem__(slice(grid_line + 40, None)))):
        lines = list(nh)
        edit(lines)
        p = tmp_path / name
        p.write_text("\n".join(lines))
        assert L.sa_model_load(C.byref(h), cases.MODEL_R73.encode(), str(p).encode()) != 0 and not h.value, name


def test_ring_kernel_routing_and_path_records(oracle):
    """Regions with ambiguous positions, and one-path regions whose band is mostly wider than a wave, are planned for the
    LDS-ring kernels; the per-path neighbour records the planner writes for the former equal path_checkLegal
    (impl/pairwiseAligner.c:595-621) evaluated pair by pair, for two- and three-letter ambiguity codes, runs of adjacent
    ambiguous letters (up to 3^3 * 2 paths in a window) and the default table."""
    p = sa.default_params()
    # (a) dense anchors, one path per cell: register kernels
    pm6 = sa.Model.load(cases.MODEL_6MER)
    dense = cases.synthetic_jobs(cases.MODEL_6MER, 3, 900, 5)
    info, _ = sa.plan_digest(pm6, p, dense)
    assert info.n_fast_regions == info.n_regions == 3 and info.n_ring_regions == 0
    # (b) anchors as sparse as a real guide alignment: ring kernels, no per-path records
    sparse = cases.realistic_anchor_jobs(cases.MODEL_6MER, 3, 1500, 5)
    info, _ = sa.plan_digest(pm6, p, sparse)
    assert info.n_ring_regions == info.n_regions == 3 and info.n_fast_regions == 0
    assert sa.plan_check_path_records(pm6, p, sparse[0]) == (0, 0)
    info, _ = sa.plan_digest(pm6, p, sparse, flags=sa.FLAG_EXACT)
    assert info.n_ring_regions == 0
    # (c) CpG model, every CpG cytosine X -> C/E
    pmc = sa.Model.load(cases.MODEL_CPG)
    amb = sa.default_ambig({"X": "CE"})
    cpg = cases.synthetic_jobs(cases.MODEL_CPG, 3, 700, 11, cpg_ambiguous=True)
    info, _ = sa.plan_digest(pmc, p, cpg, ambig=amb)
    assert info.n_ring_regions == info.n_regions == 3
    for job in cpg:
        bad, seen = sa.plan_check_path_records(pmc, p, job, ambig=amb)
        assert bad == 0 and seen > len(job["ref"])
    # the thread count does not change what is planned (records included in the digest)
    _, d1 = sa.plan_digest(pmc, p, cpg * 4, ambig=amb, threads=1)
    _, d3 = sa.plan_digest(pmc, p, cpg * 4, ambig=amb, threads=3)
    assert d1 == d3
    # (d) R7.3 ACEGOT model with the default table: L -> C/E/O, P -> C/E, adjacent and clustered
    pm7 = sa.Model.load(cases.MODEL_R73)
    job = dict(cases.synthetic_jobs(cases.MODEL_R73, 1, 400, 3)[0])
    ref = list(job["ref"])
    for i in (10, 11, 12, 40, 42, 44, 45, 100, 150, 151, 200):
        ref[i] = "L" if i % 2 == 0 else "P"
    job["ref"] = "".join(ref)
    bad, seen = sa.plan_check_path_records(pm7, p, job)
    assert bad == 0 and seen > len(ref)
    # (e) an ambiguity code with a repeated option: the index form of legality does not hold -> memory-resident kernels
    info, _ = sa.plan_digest(pmc, p, cpg, ambig=sa.default_ambig({"X": "CC"}))
    assert info.n_ring_regions == 0


def test_two_distribution_emission_keeps_one_path_regions_on_the_register_kernels():
    # round 6: the two-distribution emissions exist in the register kernels (k_fwd_fast_two / k_bwd_fast_two) and in the
    # reference-ordered ones, not in the strip / ring kernels: the planner leaves a one-path region with a wide band a register-kernel
    # region (their in-kernel memory-resident path), and a region with several paths per cell is no register-kernel region at all
    # (sa_batch_create then plans the batch again as with SA_FLAG_EXACT)
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    p = sa.default_params()
    wide = cases.realistic_anchor_jobs(cases.MODEL_6MER, 1, 1500, 31)[0]
    amb = dict(cases.synthetic_jobs(cases.MODEL_6MER, 1, 600, 32)[0])
    amb["ref"] = amb["ref"][:100] + "X" + amb["ref"][101:]
    for emission in (0, 1, 2):
        m = sa.Model.create(alpha, k, t10, tab)
        m.set_emission(emission)
        info = sa.plan_describe(m, p, wide)[0]
        if emission == 0:
            assert info.n_ring_regions == info.n_regions >= 1 and info.n_fast_regions == 0      # the strip kernels' region
        else:
            assert info.n_fast_regions == info.n_regions >= 1 and info.n_ring_regions == 0
        info = sa.plan_describe(m, p, amb, ambig=sa.default_ambig({"X": "CT"}))[0]
        assert info.n_fast_regions == 0 and (info.n_ring_regions == info.n_regions) == (emission == 0)
        m.close()
