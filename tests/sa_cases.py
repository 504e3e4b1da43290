"""Shared test inputs: the same seeded jobs are handed to the CPU oracle and to the HIP library."""
import os

import numpy as np

from signalalign_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODEL_6MER = os.path.join(GOLDEN, "models", "testModelR9.4_450bps.nucleotide.6mer.template.model")
MODEL_CPG = os.path.join(GOLDEN, "models", "testModelR9.4_450bps.cpg.6mer.template.model")
MODEL_5MER = os.path.join(GOLDEN, "models", "testModelR9_5mer_acgt_template.model")
MODEL_R73 = os.path.join(GOLDEN, "models", "testModelR73_acegot_template.model")
NHDP = os.path.join(GOLDEN, "models", "templateSingleLevelFixed.nhdp")


def synthetic_jobs(model_path, n_reads, n_events, first_index=0, **kw):
    alpha, k, t10, tab = synth.parse_model_table(model_path)
    return synth.make_jobs(n_reads, n_events, alpha, k, tab, first_index=first_index, **kw)


def realistic_anchor_jobs(model_path, n_reads, n_events, first_index=0, trim=14):
    """Synthetic reads whose anchors are thinned the way a real guide alignment thins them: match runs taken from
    the bundled minus-strand cigar (tests/golden/cigars), `trim` anchors dropped at both ends of each run, nothing
    kept inside deletions.  About one base in six stays an anchor and most diagonals are wider than 64 lanes."""
    toks = open(os.path.join(GOLDEN, "cigars", "ecoli_minus_strand.cigar")).read().split()[10:]
    runs = [(toks[i], int(toks[i + 1])) for i in range(0, len(toks), 2)]
    jobs = []
    for j, job in enumerate(synthetic_jobs(model_path, n_reads, n_events, first_index)):
        ax, ay = job["ax"], job["ay"]
        keep = np.zeros(len(ax), dtype=bool)
        pos, r = 0, (7 * j) % len(runs)
        while pos < len(ax):
            op, ln = runs[r % len(runs)]
            r += 1
            if op == "M":
                if ln > 2 * trim:
                    keep[pos + trim: min(pos + ln - trim, len(ax))] = True
                pos += ln
            elif op == "D":
                pos += ln
        q = dict(job)
        q["ax"], q["ay"] = ax[keep], ay[keep]
        jobs.append(q)
    return jobs


def oracle_pairs(oracle, omodel, job, params, ambig=None):
    omodel.set_read_params(job["scale"], job["shift"], job["var"])
    return oracle.align(omodel, job["ref"], job["events"], job["ax"], job["ay"], params, ambig=ambig)


def oracle_params(oracle, p):
    """translate a signalalign_amd.Params into the oracle's Params"""
    return oracle.Params(p.threshold, p.diagonal_expansion, p.trace_back_diagonals, p.min_diags_between_trace_back,
                         p.split_matrix_bigger_than_this, 14)


def npread_job(oracle, npread_name, model_path, reference=None):
    """A bundled .npRead aligned to `reference` (default: its own template read) with a single-M guide
    alignment: the substitution for BASELINE config 1 described in SURVEY.md section 8(c)."""
    r = oracle.parse_npread(os.path.join(GOLDEN, "npReads", npread_name))
    om = oracle.Model.from_file(model_path)
    ev = r["template_events"].copy()
    read = r["template_read"]
    pr = oracle.estimate_params(om, r["template_strand_event_map"], ev, read)
    ref = reference if reference is not None else read
    L = min(len(ref), len(read))
    gx, gy = oracle.guide_to_anchors(0, L, 1, 0, [(0, L)], 14)
    em = r["template_strand_event_map"]
    ax, ay = oracle.remap_anchors(gx, gy, em, 0)
    lo, hi = int(em[0]), int(em[L - 1])
    return dict(ref=ref[:L], events=np.ascontiguousarray(ev[lo:hi]), ax=ax, ay=ay, scale=pr["scale"],
                shift=pr["shift"], var=pr["var"])


def compare_pairs(got, exp, tol_e7, threshold):
    """got/exp: structured arrays.  Rows must agree on (x, y, path) and differ by at most tol_e7 in prob_e7,
    except rows whose probability is within tol of the threshold (they may appear on one side only).
    Returns (max_abs_diff_e7, n_only_one_side)."""
    def key(a):
        return {(int(r["x"]), int(r["y"]), int(r["path"])): int(r["prob_e7"]) for r in a}
    g, e = key(got), key(exp)
    assert len(g) == len(got) and len(e) == len(exp), "duplicate (x,y,path) rows"
    worst, lonely = 0, 0
    thr_e7 = threshold * 1e7
    for k, v in e.items():
        if k in g:
            worst = max(worst, abs(g[k] - v))
        else:
            assert abs(v - thr_e7) <= tol_e7 + 1, ("missing pair", k, v)
            lonely += 1
    for k, v in g.items():
        if k not in e:
            assert abs(v - thr_e7) <= tol_e7 + 1, ("extra pair", k, v)
            lonely += 1
    assert worst <= tol_e7, ("posterior differs by more than the tolerance", worst)
    return worst, lonely


def same_order(got, exp):
    """rows present on both sides appear in the same relative order"""
    ke = [(int(r["x"]), int(r["y"]), int(r["path"])) for r in exp]
    kg = [(int(r["x"]), int(r["y"]), int(r["path"])) for r in got]
    se, sg = set(ke), set(kg)
    return [k for k in ke if k in sg] == [k for k in kg if k in se]
