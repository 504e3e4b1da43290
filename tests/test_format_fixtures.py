"""The reference's own output files (tests/golden/format, committed as data) against the format checkers the GPU CLI tests
apply to the product's files (tests/format_checks.py), and the reference's literal build_kmer_list vectors for the event
alignment row."""
import gzip
import os

import numpy as np

import format_checks as fc
import sa_cases as cases

FMT = os.path.join(cases.GOLDEN, "format")


def test_golden_full_tsv_passes_the_checker():
    text = open(os.path.join(FMT, "zymo_C_sm3_7f22f937.forward.t300_c300.tsv")).read()
    rows = fc.check_full_rows(text, 6, "ACEGOT")
    assert len(rows) == 600 and {r[4] for r in rows} == {"t", "c"} and {r[0] for r in rows} == {"ZYMO"}


def test_golden_assignments_pass_the_checker():
    text = open(os.path.join(FMT, "d6160b0b-a35e-43b5-947f-adaa1abade28.sm.assignments.head500.tsv")).read()
    assert len(fc.check_assignment_rows(text, 6, "ACGT")) == 500


def test_golden_expectations_file_passes_the_checker():
    # tests/test_expectation_files/4f9a316c-...: ACEGT 6-mer -> 4 / 10 / 78125 / 31250 / 15625 / 15625 tokens
    text = gzip.open(os.path.join(FMT, "4f9a316c-8bb3-410a-8cfc-026061f7e8db.template.expectations.tsv.gz"), "rt").read()
    lines = fc.check_expectations_file(text, 5, "ACEGT", 6)
    assert [len([t for t in l.split("\t") if t != ""]) for l in lines[:6]] == [4, 10, 78125, 31250, 15625, 15625]
    # gapX->gapY and gapY->gapX sit at the pseudocount: the two transitions no live path uses
    t = lines[1].split("\t")
    assert t[5] == "0.001000" and t[7] == "0.001000"


def test_build_kmer_list_reference_vectors(oracle):
    # tests/eventAlignerTests.c:582-598 test_build_kmer_list: DNA ATGCATGC -> ATGCA TGCAT GCATG CATGC;
    # RNA AUGCAUGC -> ACGTA TACGT GTACG CGTAC (U read as T, every k-mer reversed)
    import signalalign_amd as sa
    om = oracle.Model.from_file(cases.MODEL_5MER)
    dna = [om.kmer_id(s) for s in ("ATGCA", "TGCAT", "GCATG", "CATGC")]
    rna = [om.kmer_id(s) for s in ("ACGTA", "TACGT", "GTACG", "CGTAC")]
    assert oracle.kmer_ids_of(om, "ATGCATGC").tolist() == dna
    assert oracle.kmer_ids_of(om, "AUGCAUGC", rna=True).tolist() == rna
    # the product's k-mer walk (sa_scalings_mom reads exactly these k-mers): identical moments from the literal lists
    pm = sa.Model.load(cases.MODEL_5MER)
    ev = np.array([80.0, 95.5, 101.25, 77.0, 88.0, 110.0], dtype=np.float64)
    for seq, ids, flags in (("ATGCATGC", dna, 0), ("AUGCAUGC", rna, sa.FLAG_RNA)):
        sh, sc = sa.scalings_mom(pm, seq, ev, flags=flags)
        osh, osc = oracle.scalings_mom(om, ev, np.array(ids, dtype=np.int32))
        assert sh == osh and sc == osc
    assert sa.scalings_mom(pm, "AUGCAUGC", ev, flags=sa.FLAG_RNA) != sa.scalings_mom(pm, "ATGCATGC", ev)


def test_golden_rna_outputs_fix_the_strand_conventions():
    """The reference's own RNA outputs (tests/test_variantCalled_files/rna/, first 150 rows each, against
    tests/test_sequences/fake_rna_ref.fa) state the conventions `signalMachine --rna` must keep (tests/test_gpu_cli.py checks the
    product's files against exactly these relations):
      forward file   positions FALL as the events advance; the target k-mer is the forward reference REVERSED in place; the
                     reference k-mer column repeats it;
      backward file  positions GROW; the target k-mer is the COMPLEMENT of the forward reference in place (not reversed); the
                     reference k-mer column is its reverse complement."""
    ref = "".join(open(os.path.join(cases.GOLDEN, "sequences", "fake_rna_ref.fa")).read().split("\n")[1:])
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    fwd = fc.check_full_rows(open(os.path.join(FMT, "rna_7d31de25.sm.forward.head150.tsv")).read(), 5, "ACGT")
    bwd = fc.check_full_rows(open(os.path.join(FMT, "rna_8898d755.sm.backward.head150.tsv")).read(), 5, "ACGT")
    assert len(fwd) == 150 and len(bwd) == 150
    for r in fwd:
        pos = int(r[1])
        assert r[9] == ref[pos:pos + 5][::-1] and r[2] == r[9] and r[4] == "t"
    for r in bwd:
        pos = int(r[1])
        assert r[9] == "".join(comp[c] for c in ref[pos:pos + 5]) and r[4] == "t"
        assert r[2] == "".join(comp[c] for c in reversed(r[9]))
    ev_f, pos_f = np.array([int(r[5]) for r in fwd]), np.array([int(r[1]) for r in fwd])
    ev_b, pos_b = np.array([int(r[5]) for r in bwd]), np.array([int(r[1]) for r in bwd])
    # (events stall and jump along a real read: the sign of the trend is what the conventions fix)
    assert np.corrcoef(ev_f, pos_f)[0, 1] < -0.7 and np.corrcoef(ev_b, pos_b)[0, 1] > 0.7
    assert pos_f[0] > pos_f[-1] and pos_b[0] < pos_b[-1]
