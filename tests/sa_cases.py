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
    return thin_anchors_like_a_guide_alignment(synthetic_jobs(model_path, n_reads, n_events, first_index), trim)


def thin_anchors_like_a_guide_alignment(job_list, trim=14):
    """the thinning of realistic_anchor_jobs for any list of jobs (HDP reads too)"""
    toks = open(os.path.join(GOLDEN, "cigars", "ecoli_minus_strand.cigar")).read().split()[10:]
    runs = [(toks[i], int(toks[i + 1])) for i in range(0, len(toks), 2)]
    jobs = []
    for j, job in enumerate(job_list):
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
    return oracle.align(omodel, job["ref"], job["events"], job["ax"], job["ay"], params, ambig=ambig, ragged=job.get("ragged", (1, 1)))


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


_HDP_CACHE = {}


def hdp_jobs(n_reads, n_events, first_index=0, table5=None):
    """Reads of the HDP workload (BASELINE configs[3]): reference assembled from windows of the sequence the bundled .nhdp was
    trained on, events drawn from the density the aligner itself uses for each k-mer (synth.make_read_hdp).  table5: the
    model's table after set_to_hdp_expected_values (sa.Model.table5() or the oracle's match_table())."""
    if "sampler" not in _HDP_CACHE:
        _HDP_CACHE["sampler"] = synth.HdpSampler(synth.parse_nhdp(NHDP))
        _HDP_CACHE["pool"] = open(os.path.join(GOLDEN, "npReads", "ZymoRef.txt")).read().split()[0].strip()
    alpha, k, t10, tab = synth.parse_model_table(MODEL_R73)
    t5 = tab if table5 is None else np.asarray(table5)
    return [synth.make_read_hdp(first_index + i, n_events, alpha, k, t5, _HDP_CACHE["sampler"], _HDP_CACHE["pool"])
            for i in range(n_reads)]


def events_for_sequence(seq, model_path, seed):
    """Synthetic events for a given nucleotide sequence (as synth.make_read draws them for a random one): 0-7 events per
    k-mer, means ~ N(mu_k, sd_k).  Returns (events4, event_map) with event_map[i] = first event of the k-mer at base i."""
    alpha, k, t10, tab = synth.parse_model_table(model_path)
    alpha = "".join(sorted(alpha))
    rng = np.random.Generator(np.random.PCG64(seed))
    n_kmers = len(seq) - k + 1
    digit = np.array([alpha.index(c) for c in seq], dtype=np.int64)
    kid = np.zeros(n_kmers, dtype=np.int64)
    for i in range(k):
        kid = kid * len(alpha) + digit[i:i + n_kmers]
    p = synth.EVENTS_PER_KMER_P / synth.EVENTS_PER_KMER_P.sum()
    counts = rng.choice(len(p), size=n_kmers, p=p)
    counts[0] = max(counts[0], 1)
    owner = np.repeat(np.arange(n_kmers), counts)
    E = len(owner)
    means = rng.normal(tab[5 * kid[owner]], tab[5 * kid[owner] + 1])
    noise = np.abs(rng.normal(tab[5 * kid[owner] + 2], tab[5 * kid[owner] + 3])) + 1e-3
    dur = np.full(E, 0.00127)
    events4 = np.ascontiguousarray(np.stack([means, noise, dur, np.cumsum(dur) - dur], axis=1))
    first = np.cumsum(counts) - counts
    idx = np.maximum.accumulate(np.where(counts > 0, np.arange(n_kmers), 0))
    emap = np.full(len(seq), E - 1, dtype=np.int64)
    emap[:n_kmers] = np.minimum(first[idx], E - 1)
    return events4, emap


def write_npread_1d(path, read, event_map, events4):
    """A 1-D .npRead as nanopore_loadNanoporeReadFromFile reads it (impl/nanopore.c:145-521; layout of the bundled
    tests/test_npReads/r9p4_oneD.npRead): header, empty 2-D read, template read, template strand event map, three empty
    complement lines, the template events, empty lines."""
    with open(path, "w") as f:
        f.write("0 %d 0 %d 0 1 1 1 1 1 0 1 1 1 1 1 0 0\n" % (len(events4), len(read)))
        f.write("\n%s\n%s\n\n\n\n" % (read, " ".join(str(int(v)) for v in event_map)))
        f.write(" ".join(repr(float(v)) for v in np.asarray(events4).reshape(-1)) + "\n")
        f.write("\n\n\n\n\n\n\n")


def reference_output_ecoli1d_inputs(oracle):
    """The reference's output for the bundled R9.4 1-D read (tests/golden/expected/reference_output_ecoli1d.npz) as an alignment
    job: (posteriors by (window position, event), reference window rebuilt from the rows' k-mers, parsed .npRead, (start1, end1,
    start2, end2), guide-alignment operations).  The E. coli genome is a missing blob and the guide alignment bwa made is not
    shipped: the operations are rebuilt from the rows themselves -- per reference position its most probable event (p >= 0.5)
    mapped to a read base -- which is cruder than the original."""
    z = np.load(os.path.join(GOLDEN, "expected", "reference_output_ecoli1d.npz"))
    gold = {(int(x), int(y)): float(p) for x, y, p in zip(z["x"], z["y"], z["p"])}
    window = str(z["window"]).replace("?", "A")      # (one base no row covers)
    r = oracle.parse_npread(os.path.join(GOLDEN, "npReads", "r9p4_oneD.npRead"))
    em = r["template_strand_event_map"]
    best = {}
    for (x, y), p in gold.items():
        if p >= 0.5 and (x not in best or p > best[x][1]):
            best[x] = (y, p)
    m, last = [], -1
    for x in sorted(best):
        b = int(np.searchsorted(em, best[x][0], side="right") - 1)
        if b > last:
            m.append((x, b))
            last = b
    ops = []

    def push(t, n):
        if n > 0:
            if ops and ops[-1][0] == t:
                ops[-1] = (t, ops[-1][1] + n)
            else:
                ops.append((t, n))
    for (x, b), (x2, b2) in zip(m[:-1], m[1:]):
        mm = min(x2 - x, b2 - b)
        push(0, mm)
        push(1, x2 - x - mm)
        push(2, b2 - b - mm)
    push(0, 1)
    return gold, window, r, (m[0][0], m[-1][0] + 1, m[0][1], m[-1][1] + 1), ops


def reference_residual(mine, gold):
    """How a run's posteriors compare with the ones the reference printed (round 4, probes/reference_output_residuals.py).
    The rows that differ by more than 1e-4 do not sit at band edges: they come in groups of consecutive diagonals that share ONE
    multiplicative factor, |log factor| <= 1.5e-3 -- the total probability a posterior is divided by.  The reference refreshes
    that total on every tenth diagonal of a traceback, and its approximate logAdd makes the total depend on the diagonal it is
    evaluated on (un-banded restatement on the Zymo read: totals ten diagonals apart differ by 6e-5 in the median, 5e-4 at the
    95th percentile, 1.7e-3 at most); which diagonals those are depends on where the tracebacks fire, i.e. on the guide
    alignment, which cannot be the reference's (bwa's is not shipped).  So the bar that the explanation supports is RELATIVE:
    |dp| <= 1.5e-3 p + 2e-6 (the print precision).  Returns (share of the reference's rows found, median |dp|, share of the found
    rows within that bar, share within 1e-4 absolute, the rows beyond the bar as (x + y, x, y, reference p, this run's p))."""
    common = sorted(set(mine) & set(gold))
    g = np.array([gold[k] for k in common])
    m = np.array([mine[k] for k in common])
    d = np.abs(m - g)
    ok = d <= 1.5e-3 * np.maximum(g, m) + 2e-6
    beyond = [(k[-2] + k[-1], k[-2], k[-1], float(gg), float(mm)) for k, gg, mm, o in zip(common, g, m, ok) if not o]
    return len(common) / max(len(gold), 1), float(np.median(d)), float(ok.mean()), float((d <= 1e-4).mean()), beyond
