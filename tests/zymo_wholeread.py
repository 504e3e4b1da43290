"""The reference's whole-read known answers on ZymoC_ch_1_file1.npRead x ZymoRef.txt (tests/stateMachineTests.c:842-983).

Those tests take their anchors from a cPecanLastz subprocess (getRemappedAnchors, tests/stateMachineTests.c:130-137).
The raw cigar lines of that subprocess are committed as data (tests/golden/cigars/zymoC_lastz_anchors.json, written by
tests/golden/make_lastz_cigars.py); everything after the subprocess is restated here in plain Python, independent of
both the oracle's and the product's C code:

* cigar_to_pairs   = convertPairwiseForwardStrandAlignmentToAnchorPairs  impl/pairwiseAligner.c:1624-1658
* filter_overlap   = filterToRemoveOverlap                               impl/pairwiseAligner.c:1755-1796
* top_level_anchors= getBlastPairsForPairwiseAlignmentParameters         impl/pairwiseAligner.c:1826-1877
* remapped_anchors = nanopore_remapAnchorPairs + filterToRemoveOverlap   impl/nanopore.c:523-533
"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRIM = 14                      # constraintDiagonalTrim of pairwiseAlignmentBandingParameters_construct, :2028
ANCHOR_MATRIX = 500 * 500      # anchorMatrixBiggerThanThis == repeatMaskMatrixBiggerThanThis, :2030-2031
# pairwiseAlignmentBandingParameters_construct (impl/pairwiseAligner.c:2022-2037): threshold 0.01, diagonalExpansion 20,
# traceBackDiagonals 40, minDiagsBetweenTraceBack 1000, splitMatrixBiggerThanThis 3000*3000
BANDING = dict(threshold=0.01, expansion=20, trace_back=40, min_diags=1000, split=3000 * 3000)
# expected list lengths: tests/stateMachineTests.c:851-852, :863, :912, :941-970
N_PAIRS_GAUSS = 1076
N_PAIRS_HDP_AT_0p1 = 1217
N_PAIRS_DEGENERATE = {"C": 1076, "E": 1076, "O": 1076, "L": 7349}


def cigar_to_pairs(line, trim=TRIM):
    """sonLib cigarRead order: 'cigar: <contig2> <start2> <end2> <strand2> <contig1> <start1> <end1> <strand1> <score>'
    then (op, length)*: M advances both, D only sequence 1 (PAIRWISE_INDEL_X), I only sequence 2."""
    t = line.split()
    assert t[0] == "cigar:" and t[4] == "+" and t[8] == "+", line
    start2, end2, start1, end1 = int(t[2]), int(t[3]), int(t[6]), int(t[7])
    ops = t[10:]
    j, k, out = start1, start2, []
    for i in range(0, len(ops), 2):
        op, n = ops[i], int(ops[i + 1])
        if op == "M":
            for l in range(trim, n - trim):
                if end1 >= j + l + 6:
                    out.append((j + l, k + l))
        if op != "I":
            j += n
        if op != "D":
            k += n
    assert j == end1 and k == end2, (line, j, k)
    return out


def filter_overlap(pairs):
    keep = set()
    px = py = float("inf")
    for x, y in reversed(pairs):
        if x < px and y < py:
            keep.add((x, y))
        px, py = min(px, x), min(py, y)
    out = []
    px = py = float("-inf")
    for x, y in pairs:
        if x > px and y > py and (x, y) in keep:
            out.append((x, y))
        px, py = max(px, x), max(py, y)
    return out


def blast_pairs_from_cigars(lines, trim=TRIM):
    """getBlastPairs after the pipe (:1713-1727) + the caller's sort and filter (:1841-1845)"""
    pairs = []
    for l in lines:
        pairs += cigar_to_pairs(l, trim)
    pairs.sort(key=lambda p: p[0] + p[1])   # sortByXPlusYCoordinate (stable, as glibc's merge sort is)
    pairs.sort()                             # stIntTuple_cmpFn
    return filter_overlap(pairs)


def top_level_anchors():
    rec = json.load(open(os.path.join(GOLDEN, "cigars", "zymoC_lastz_anchors.json")))
    calls = {(c["pX"], c["pY"]): c for c in rec["calls"]}
    first = calls[(0, 0)]
    lX, lY = first["lX"], first["lY"]
    assert lX * lY > ANCHOR_MATRIX
    top = blast_pairs_from_cigars(first["cigars"])
    combined, used = [], 1
    px = py = 0
    for (x, y) in top + [(lX, lY)]:
        if (x - px) * (y - py) > ANCHOR_MATRIX:          # getBlastPairsForPairwiseAlignmentParametersP :1798-1824
            c = calls[(px, py)]                           # KeyError = the committed file lacks a call the recursion makes
            assert (c["lX"], c["lY"]) == (x - px, y - py)
            combined += [(a + px, b + py) for a, b in blast_pairs_from_cigars(c["cigars"])]
            used += 1
        if (x, y) != (lX, lY):
            combined.append((x, y))
        px, py = x + 1, y + 1
    assert used == len(calls)
    return combined, lX, lY


def read_fixture():
    """Plain-text view of the fixture (impl/nanopore.c:145-521 line layout), no C code involved."""
    lines = open(os.path.join(GOLDEN, "npReads", "ZymoC_ch_1_file1.npRead")).read().split("\n")
    h = lines[0].split()
    keys = ["scale", "shift", "var", "scale_sd", "var_sd", "drift"]
    r = dict(twoD_read=lines[1].strip(), template_params={k: float(v) for k, v in zip(keys, h[5:11])},
             template_event_map=np.array(lines[6].split(), dtype=np.int64),
             template_events=np.array(lines[7].split(), dtype=np.float64).reshape(-1, 4))
    r["ref"] = open(os.path.join(GOLDEN, "npReads", "ZymoRef.txt")).readline().strip()
    return r


def remapped_anchors():
    """getRemappedAnchors (tests/stateMachineTests.c:130-137): anchors in (k-mer index, template event index)."""
    r = read_fixture()
    top, lX, lY = top_level_anchors()
    assert lX == len(r["ref"]) and lY == len(r["twoD_read"])
    em = r["template_event_map"]
    fil = filter_overlap([(x, int(em[y])) for x, y in top])
    return np.array([p[0] for p in fil], dtype=np.int64), np.array([p[1] for p in fil], dtype=np.int64)


def scaled_table(table5, tp):
    """emissions_signal_scaleModel (impl/stateMachine.c:743-779) on the EMISSION_MATCH_MATRIX columns"""
    t = np.array(table5, dtype=np.float64).reshape(-1, 5).copy()
    t[:, 0] = t[:, 0] * tp["scale"] + tp["shift"]
    t[:, 1] = t[:, 1] * tp["var"]
    t[:, 2] = t[:, 2] * tp["scale_sd"]
    t[:, 4] = t[:, 4] * tp["var_sd"]
    t[:, 3] = np.sqrt(np.power(t[:, 2], 3.0) / t[:, 4])
    return t.reshape(-1)


def hdp_test_events(r):
    """nanopore_descaleNanoporeRead as test_sm3Hdp_getAlignedPairsWithBanding calls it (tests/stateMachineTests.c:905):
    nanopore_descaleEvents (impl/nanopore.c:83-87) steps its index by NB_EVENT_PARAMS but stops at nb_events, so only the
    means of the first ceil(nb_events / 4) events are descaled.  The expected count 1217 holds for exactly that input."""
    tp = r["template_params"]
    ev = r["template_events"].copy()
    n4 = (ev.shape[0] + 3) // 4
    ev[:n4, 0] = (ev[:n4, 0] - tp["shift"]) / tp["scale"]
    return ev
