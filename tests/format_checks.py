"""Format checkers shared by the CPU tests (run on the reference's own golden output files, committed as data under
tests/golden/format) and the GPU CLI tests (run on what signalMachine writes here).  A checker that accepts the golden file
and the product's file pins the product's writers to the reference's format.

Golden files (copied unmodified, the first two cut to their first rows):
  * zymo_C_sm3_7f22f937.forward.t300_c300.tsv   tests/test_alignments/zymo_C_test_alignments_sm3/tempFiles_alignment/
                                                7f22f937-..._Basecall_2D_000_2d.sm.forward.tsv (300 template + 300
                                                complement rows): writePosteriorProbsFull, impl/signalMachine.c:89-159
  * d6160b0b-....sm.assignments.head500.tsv     tests/test_assignment_files/: writeAssignments, impl/signalMachine.c:234-270
  * 4f9a316c-....template.expectations.tsv.gz   tests/test_expectation_files/: continuousPairHmm_writeToFile,
                                                impl/continuousHmm.c:353-407 (gzip of the whole file)
"""
import re

import numpy as np

F6 = re.compile(r"^-?\d+\.\d{6}$")          # "%f" / "%lf"
INT = re.compile(r"^-?\d+$")


def check_full_rows(text, k, alphabet):
    """16 columns: contig, ref position, reference k-mer, read label, strand, event index, event mean/noise/duration (%f),
    target k-mer, scaled model mean, scaled model noise, posterior, descaled event mean, model mean (%f), path k-mer.
    Returns the rows split into fields.  Per (label, strand) the columns are tied by the writer's formulas
    (impl/signalMachine.c:131-141): col11 = col15 * scale + shift and col14 = (col7 - col11) / var + col15 for ONE
    (scale, shift, var) -- checked by fitting them and bounding the residual by the %f rounding."""
    rows = [l.split("\t") for l in text.split("\n") if l]
    assert rows
    kmer = re.compile("^[%s]{%d}$" % (re.escape(alphabet), k))
    groups = {}
    for r in rows:
        assert len(r) == 16, r
        assert INT.match(r[1]) and INT.match(r[5]) and r[4] in ("t", "c"), r
        for c in (6, 7, 8, 10, 11, 12, 13, 14):
            assert F6.match(r[c]), (c, r)
        for c in (2, 9, 15):
            assert kmer.match(r[c]), (c, r)
        assert 0.0 <= float(r[12]) <= 1.0
        groups.setdefault((r[3], r[4]), []).append(r)
    for key, g in groups.items():
        e_mean = np.array([float(r[14]) for r in g])
        scaled = np.array([float(r[10]) for r in g])
        if len(set(e_mean.tolist())) < 3:
            continue
        A = np.stack([e_mean, np.ones_like(e_mean)], axis=1)
        (scale, shift), *_ = np.linalg.lstsq(A, scaled, rcond=None)
        assert np.abs(A @ np.array([scale, shift]) - scaled).max() < 2e-6 * max(1.0, abs(scale)) + 1e-6, key
        ev = np.array([float(r[6]) for r in g])
        desc = np.array([float(r[13]) for r in g])
        num = ev - scaled
        den = desc - e_mean
        ok = np.abs(den) > 1.0
        if ok.sum() >= 3:
            var = np.median(num[ok] / den[ok])
            assert np.abs(num / var + e_mean - desc).max() < 1e-4, key
    return rows


def check_assignment_rows(text, k, alphabet):
    """writeAssignments (impl/signalMachine.c:234-270): k-mer, strand, descaled event mean (%lf), posterior (%lf)"""
    rows = [l.split("\t") for l in text.split("\n") if l]
    kmer = re.compile("^[%s]{%d}$" % (re.escape(alphabet), k))
    for r in rows:
        assert len(r) == 4 and kmer.match(r[0]) and r[1] in ("t", "c") and F6.match(r[2]) and F6.match(r[3]), r
        assert 0.0 <= float(r[3]) <= 1.0
    return rows


def check_expectations_file(text, n_alpha, alphabet, k):
    """continuousPairHmm_writeToFile: six lines; header, event model, expectations, posteriors and mask are tab-TERMINATED,
    the transitions line ends with the likelihood and no tab.  Token counts 4 / 10 / 5 A^k / 2 A^k / A^k / A^k."""
    lines = text.split("\n")
    assert len(lines) == 7 and lines[6] == "", len(lines)
    n = n_alpha ** k
    assert lines[0] == "3\t%d\t%s\t%d\t" % (n_alpha, alphabet, k)
    t = lines[1].split("\t")
    assert len(t) == 10 and all(F6.match(x) for x in t)
    for i, cnt in ((2, 5 * n), (3, 2 * n), (4, n)):
        f = lines[i].split("\t")
        assert len(f) == cnt + 1 and f[-1] == "", (i, len(f))
        assert all(F6.match(x) for x in f[:200]) and all(F6.match(x) for x in f[-200:-1])
    m = lines[5].split("\t")
    assert len(m) == n + 1 and m[-1] == "" and set(m[:-1]) <= {"0", "1"}
    return lines
