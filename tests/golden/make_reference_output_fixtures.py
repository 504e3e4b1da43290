"""Writes tests/golden/expected/reference_output_*.npz from two output files the reference ships (run in the build container,
where /root/reference exists; the fixtures are data: positions, event indices, posteriors and the reference k-mers of the rows).

  zymo2d   tests/test_alignments/zymo_C_test_alignments_sm3/tempFiles_alignment/7f22f937-...sm.forward.tsv
           = signalMachine's output for tests/test_npReads/ZymoC_ch_1_file1.npRead (2-D, R7.3, ZYMO contig)
  ecoli1d  tests/test_alignments/ecoli1D_test_alignments_sm3/6deaf971-...sm.forward.tsv
           = signalMachine's output for tests/test_npReads/r9p4_oneD.npRead (1-D, R9.4 5-mer ACEGT model, E. coli window).
           The E. coli reference is a missing blob; the window the read aligns to is rebuilt from the rows' own k-mers.
"""
import os

import numpy as np

REF = "/root/reference/tests/test_alignments"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "expected")


def rows_of(path):
    return [l.rstrip("\n").split("\t") for l in open(path)]


def main():
    z = rows_of(os.path.join(REF, "zymo_C_test_alignments_sm3", "tempFiles_alignment",
                             "7f22f937-c340-4cec-a099-d3ac14a448c0_Basecall_2D_000_2d.sm.forward.tsv"))
    np.savez_compressed(os.path.join(OUT, "reference_output_zymo2d.npz"),
                        strand=np.array([g[4] for g in z]), x=np.array([int(g[1]) for g in z], dtype=np.int32),
                        y=np.array([int(g[5]) for g in z], dtype=np.int32), p=np.array([float(g[12]) for g in z]))
    e = rows_of(os.path.join(REF, "ecoli1D_test_alignments_sm3", "6deaf971-6506-4e37-b486-cdf5e9d416ac.sm.forward.tsv"))
    pos = np.array([int(g[1]) for g in e])
    p0, p1, k = int(pos.min()), int(pos.max()), len(e[0][2])
    ref = ["?"] * (p1 - p0 + k)
    for g in e:
        for i, ch in enumerate(g[2]):
            assert ref[int(g[1]) - p0 + i] in ("?", ch)
            ref[int(g[1]) - p0 + i] = ch
    np.savez_compressed(os.path.join(OUT, "reference_output_ecoli1d.npz"), x=(pos - p0).astype(np.int32),
                        y=np.array([int(g[5]) for g in e], dtype=np.int32), p=np.array([float(g[12]) for g in e]),
                        window="".join(ref), first_position=p0)


if __name__ == "__main__":
    main()
