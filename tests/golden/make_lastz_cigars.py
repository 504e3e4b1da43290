"""Generates tests/golden/cigars/zymoC_lastz_anchors.json: the raw cigar lines the reference's whole-read tests obtain
from their cPecanLastz subprocess (tests/stateMachineTests.c:130-137 -> getBlastPairsForPairwiseAlignmentParameters,
impl/pairwiseAligner.c:1826-1877 -> getBlastPairs :1660-1740) for ZymoRef.txt x the 2D read of ZymoC_ch_1_file1.npRead.

Run in the build container only (needs oracle/_ref/cPecanLastz, built by oracle/build_lastz.sh from the lastz sources
the reference vendors).  lastz is an input generator, not the oracle: only its output is committed.  Every lastz
invocation of the recursion is recorded as {pX, pY, lX, lY, cigars}; tests/test_oracle_kats.py re-derives the anchors
from these lines with its own restatement of the conversion and checks that the recorded invocations are exactly the
ones the recursion asks for.
"""
import json
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zymo_wholeread as z  # noqa: E402  (the anchor conversion restated in tests/zymo_wholeread.py)

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LASTZ = os.path.join(ROOT, "oracle", "_ref", "cPecanLastz")
OPTS = ["--hspthresh=1800", "--chain", "--strand=plus", "--gapped", "--format=cigar", "--gap=100,100",
        "--ambiguous=iupac,100,100"]                       # impl/pairwiseAligner.c:1694-1703
TRIM = 14                                                   # constraintDiagonalTrim, :2028
ANCHOR_MATRIX = 500 * 500                                   # anchorMatrixBiggerThanThis / repeatMask..., :2030-2031


def run_lastz(sx, sy):
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "a.fa")
        with open(fa, "w") as f:
            f.write(">a\n%s\n" % sx)
        if len(sy) > 1000:
            fb = os.path.join(d, "b.fa")
            with open(fb, "w") as f:
                f.write(">b\n%s\n" % sy)
            out = subprocess.run([LASTZ] + OPTS + [fa, fb], check=True, capture_output=True, text=True).stdout
        else:
            out = subprocess.run([LASTZ] + OPTS + [fa], input=">b\n%s\n\n" % sy, check=True, capture_output=True,
                                 text=True).stdout
    return [l for l in out.splitlines() if l.startswith("cigar:")]


def blast_pairs(sx, sy, calls, px, py):
    lines = run_lastz(sx, sy)
    calls.append(dict(pX=px, pY=py, lX=len(sx), lY=len(sy), cigars=lines))
    return z.blast_pairs_from_cigars(lines, TRIM)


def main():
    npread = open(os.path.join(HERE, "npReads", "ZymoC_ch_1_file1.npRead")).readlines()
    two_d = npread[1].strip()
    ref = open(os.path.join(HERE, "npReads", "ZymoRef.txt")).readline().strip()
    calls = []
    assert len(ref) * len(two_d) > ANCHOR_MATRIX
    top = blast_pairs(ref, two_d, calls, 0, 0)
    px = py = 0
    for (x, y) in top + [(len(ref), len(two_d))]:
        if (x - px) * (y - py) > ANCHOR_MATRIX:
            blast_pairs(ref[px:x], two_d[py:y], calls, px, py)
        px, py = x + 1, y + 1
    out = dict(source="cPecanLastz (lastz 1.03.54 vendored by signalAlign) " + " ".join(OPTS),
               target="tests/golden/npReads/ZymoRef.txt", query="2D read (line 2) of ZymoC_ch_1_file1.npRead",
               calls=calls)
    os.makedirs(os.path.join(HERE, "cigars"), exist_ok=True)
    with open(os.path.join(HERE, "cigars", "zymoC_lastz_anchors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("lastz calls:", len(calls), "top-level anchors:", len(top), file=sys.stderr)


if __name__ == "__main__":
    main()
