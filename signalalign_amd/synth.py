"""Deterministic synthetic nanopore reads for the benchmark and the parity tests.

Follows the recipe in SURVEY.md section 8(d): i.i.d. ACGT reference, events per k-mer drawn from the
empirical distribution measured on tests/test_npReads/r9p4_oneD.npRead (1.67 events per base), event
mean ~ N(scale*mu_k + shift, (var*sd_k)^2), per-read scale ~ U(0.95,1.05), shift ~ U(-3,3), var ~ U(0.9,1.3),
guide alignment = one match run over the whole read.  Everything is produced with numpy's PCG64 seeded by
0x5A11C0DE + read_index, so the oracle and the GPU path see bit-identical inputs.

This module only makes DATA; it performs none of the aligner's arithmetic.
"""
import numpy as np

SEED0 = 0x5A11C0DE
# P(number of events for one k-mer = 0..7)
EVENTS_PER_KMER_P = np.array([0.016, 0.573, 0.252, 0.098, 0.035, 0.016, 0.006, 0.004])


def parse_model_table(path):
    """(alphabet, k, transitions10, table5) of a .model file (three whitespace-split lines)."""
    with open(path) as f:
        l0, l1, l2 = f.readline().split(), f.readline().split(), f.readline().split()
    return l0[2], int(l0[3]), np.array(l1, dtype=np.float64), np.array(l2, dtype=np.float64)


def single_match_anchors(event_map, read_len, trim=14):
    """Anchors of a guide alignment that is one match run over the whole read, remapped to events and
    filtered to be strictly increasing (what the aligner's host code produces for `cigar: ... M <len>`)."""
    x = np.arange(trim, max(read_len - trim, trim), dtype=np.int64)
    x = x[x + 6 <= read_len]
    y = event_map[x] - event_map[0]
    if len(x) == 0:
        return x, y
    # a pair survives iff it is strictly below every later pair and strictly above every earlier one
    # (running minima from the right, running maxima from the left, over ALL pairs)
    later_min = np.minimum.accumulate(y[::-1])[::-1]
    earlier_max = np.maximum.accumulate(y)
    keep = np.ones(len(x), dtype=bool)
    keep[:-1] &= y[:-1] < later_min[1:]
    keep[1:] &= y[1:] > earlier_max[:-1]
    return x[keep], y[keep]


def make_read(index, n_events, alphabet, k, table5, trim=14, cpg_ambiguous=False, thin_anchors=0.0, cpg_every=1):
    """Returns dict(ref, events4, ax, ay, scale, shift, var, event_map, read) for read `index`.

    cpg_ambiguous: replace every C that is followed by G with 'X' (config 3: CpG cytosines ambiguous); cpg_every = n > 1: only
    every n-th of them (sparse variant positions: most cells of the read then hold one path).
    thin_anchors: fraction of the read covered by anchor-free windows (realistic guide alignments).
    """
    rng = np.random.Generator(np.random.PCG64(SEED0 + int(index)))
    canonical = "ACGT"
    n_kmers = max(int(round(n_events / 1.67)), 8)
    for attempt in range(4):
        L = n_kmers + k - 1
        bases = rng.integers(0, 4, size=L)
        counts = rng.choice(len(EVENTS_PER_KMER_P), size=n_kmers, p=EVENTS_PER_KMER_P / EVENTS_PER_KMER_P.sum())
        counts[0] = max(counts[0], 1)
        total = int(counts.sum())
        if abs(total - n_events) <= max(0.01 * n_events, 2) or attempt == 3:
            break
        n_kmers = max(int(round(n_kmers * n_events / max(total, 1))), 8)
    read = "".join(canonical[b] for b in bases)
    # k-mer ids in the model's (sorted) alphabet
    alpha = "".join(sorted(alphabet))
    digit = np.array([alpha.index(c) for c in canonical], dtype=np.int64)[bases]
    kid = np.zeros(n_kmers, dtype=np.int64)
    for i in range(k):
        kid = kid * len(alpha) + digit[i:i + n_kmers]
    mu = table5[5 * kid]
    sd = table5[5 * kid + 1]
    nmean = table5[5 * kid + 2]
    nsd = table5[5 * kid + 3]
    scale, shift, var = rng.uniform(0.95, 1.05), rng.uniform(-3.0, 3.0), rng.uniform(0.9, 1.3)
    owner = np.repeat(np.arange(n_kmers), counts)
    E = len(owner)
    means = rng.normal(scale * mu[owner] + shift, var * sd[owner])
    noise = np.abs(rng.normal(nmean[owner], nsd[owner])) + 1e-3
    dur = np.full(E, 0.00127)
    start = np.cumsum(dur) - dur
    events4 = np.ascontiguousarray(np.stack([means, noise, dur, start], axis=1))
    first = np.cumsum(counts) - counts
    emap = np.zeros(L, dtype=np.int64)
    emap[:n_kmers] = np.minimum(first, E - 1)
    emap[n_kmers:] = E - 1
    # skipped k-mers point at the previous k-mer's first event (non-decreasing map)
    skipped = counts == 0
    if skipped.any():
        idx = np.where(~skipped, np.arange(n_kmers), 0)
        idx = np.maximum.accumulate(idx)
        emap[:n_kmers] = np.minimum(first[idx], E - 1)
    ax, ay = single_match_anchors(emap, L, trim)
    if thin_anchors > 0 and len(ax) > 100:
        # drop anchors in random windows of 30..300 bases
        keep = np.ones(len(ax), dtype=bool)
        target = thin_anchors * len(ax)
        dropped = 0
        while dropped < target:
            w = int(rng.integers(30, 300))
            s = int(rng.integers(0, max(len(ax) - w, 1)))
            dropped += int(keep[s:s + w].sum())
            keep[s:s + w] = False
        ax, ay = ax[keep], ay[keep]
    lo, hi = int(emap[0]), int(emap[L - 1])
    ref = read
    if cpg_ambiguous and cpg_every <= 1:
        ref = read.replace("CG", "XG")
    elif cpg_ambiguous:
        parts = read.split("CG")
        ref = parts[0]
        for q, part in enumerate(parts[1:]):
            ref += ("XG" if q % cpg_every == 0 else "CG") + part
    return dict(ref=ref, read=read, events4=events4, events=np.ascontiguousarray(events4[lo:hi]), ax=ax, ay=ay,
                scale=scale, shift=shift, var=var, event_map=emap)


def parse_nhdp(path):
    """The slice of a serialised .nhdp (impl/hdp.c:2919-3051 writes it) the HDP read generator needs: the grid, each
    process's parent, which processes saw data (every named process and its ancestors) and their posterior predictive
    values on the grid.  Data parsing only."""
    with open(path) as f:
        rd = f.readline
        n_alpha, alphabet, k = int(rd()), rd().split()[0], int(rd())
        splines, has_data, sample_gamma = int(rd()) != 0, int(rd()) != 0, int(rd()) != 0
        num_dps = int(rd())
        dp_ids = []
        if has_data:
            rd()
            dp_ids = [int(t) for t in rd().split()]
        rd()
        g0, g1, gl = rd().split()
        rd()
        if sample_gamma:
            for _ in range(4):
                rd()
        parent = np.array([-1 if t.startswith("-") else int(t.split()[0]) for t in (rd() for _ in range(num_dps))], dtype=np.int64)
        observed = np.zeros(num_dps, dtype=bool)
        for a in dp_ids:
            while a >= 0 and not observed[a]:
                observed[a] = True
                a = parent[a]
        post = {}
        if has_data:
            for i in range(num_dps):
                line = rd()
                if observed[i]:
                    post[i] = np.array(line.split(), dtype=np.float64)
    return dict(alphabet=alphabet, k=k, grid=np.linspace(float(g0), float(g1), int(gl)), parent=parent, observed=observed,
                post=post)


class HdpSampler:
    """Draws normalised events from the posterior predictive density an HDP k-mer resolves to (its own process when it
    saw data, else its nearest ancestor that did: impl/hdp.c:2600-2602), by inverse CDF over the grid (trapezoids)."""

    def __init__(self, nhdp):
        self.h = nhdp
        self.cdf = {}

    def resolve(self, kids):
        if not hasattr(self, "_res"):
            obs, par = self.h["observed"], self.h["parent"]
            res = np.arange(len(par), dtype=np.int64)
            for _ in range(64):                      # depth of the tree at most
                todo = (res >= 0) & ~obs[np.maximum(res, 0)]
                if not todo.any():
                    break
                res[todo] = par[res[todo]]
            self._res = res
        return self._res[np.asarray(kids, dtype=np.int64)]

    def draw(self, rng, kids):
        g = self.h["grid"]
        if not hasattr(self, "_cdf"):
            ids = sorted(self.h["post"])
            self._row = np.full(len(self.h["parent"]), -1, dtype=np.int64)
            self._row[ids] = np.arange(len(ids))
            d = np.maximum(np.stack([self.h["post"][i] for i in ids]), 0.0)
            c = np.concatenate([np.zeros((len(ids), 1)), np.cumsum(0.5 * (d[:, 1:] + d[:, :-1]) * np.diff(g), axis=1)], axis=1)
            self._cdf = c / c[:, -1:]
        rows = self._cdf[self._row[self.resolve(kids)]]           # one CDF row per event
        u = rng.uniform(size=len(rows))
        hi = np.clip((rows < u[:, None]).sum(axis=1), 1, len(g) - 1)   # first grid point with cdf >= u
        lo = hi - 1
        ar = np.arange(len(rows))
        c0, c1 = rows[ar, lo], rows[ar, hi]
        t = np.where(c1 > c0, (u - c0) / np.where(c1 > c0, c1 - c0, 1.0), 0.0)
        return g[lo] + t * (g[hi] - g[lo])


def make_read_hdp(index, n_events, alphabet, k, table5, sampler, ref_pool, trim=14):
    """A read for the HDP workload (BASELINE configs[3]).  The bundled .nhdp saw the k-mers of ONE real read (351 leaf
    processes; every other k-mer resolves to the root's broad density and cannot be told from its neighbours), so the
    reference is assembled from windows of the sequence that read came from (`ref_pool`: tests/golden/npReads/ZymoRef.txt)
    and every event is drawn from the density the aligner itself uses for its k-mer; table5 is the model's table after
    set_to_hdp_expected_values (its level means enter the event normalisation, impl/stateMachine.c:541-545)."""
    rng = np.random.Generator(np.random.PCG64(SEED0 + 0x48445000 + int(index)))
    n_kmers = max(int(round(n_events / 1.67)), 8)
    for attempt in range(4):
        counts = rng.choice(len(EVENTS_PER_KMER_P), size=n_kmers, p=EVENTS_PER_KMER_P / EVENTS_PER_KMER_P.sum())
        counts[0] = max(counts[0], 1)
        total = int(counts.sum())
        if abs(total - n_events) <= max(0.01 * n_events, 2) or attempt == 3:
            break
        n_kmers = max(int(round(n_kmers * n_events / max(total, 1))), 8)
    L = n_kmers + k - 1
    parts, have = [], 0
    while have < L:
        w = int(rng.integers(150, 600))
        s0 = int(rng.integers(0, max(len(ref_pool) - w, 1)))
        parts.append(ref_pool[s0:s0 + w])
        have += len(parts[-1])
    read = "".join(parts)[:L]
    alpha = "".join(sorted(alphabet))
    digit = np.array([alpha.index(c) for c in read], dtype=np.int64)
    kid = np.zeros(n_kmers, dtype=np.int64)
    for i in range(k):
        kid = kid * len(alpha) + digit[i:i + n_kmers]
    scale, shift, var = rng.uniform(0.95, 1.05), rng.uniform(-3.0, 3.0), rng.uniform(0.9, 1.3)
    owner = np.repeat(np.arange(n_kmers), counts)
    E = len(owner)
    mu = table5[5 * kid[owner]]
    en = sampler.draw(rng, kid[owner])
    means = var * en - (var - scale) * mu + shift    # inverse of e' = (e + var mu - scale mu - shift) / var
    noise = np.full(E, 1.0)
    dur = np.full(E, 0.00127)
    start = np.cumsum(dur) - dur
    events4 = np.ascontiguousarray(np.stack([means, noise, dur, start], axis=1))
    first = np.cumsum(counts) - counts
    idx = np.maximum.accumulate(np.where(counts > 0, np.arange(n_kmers), 0))
    emap = np.full(L, E - 1, dtype=np.int64)
    emap[:n_kmers] = np.minimum(first[idx], E - 1)
    ax, ay = single_match_anchors(emap, L, trim)
    lo, hi = int(emap[0]), int(emap[L - 1])
    return dict(ref=read, read=read, events4=events4, events=np.ascontiguousarray(events4[lo:hi]), ax=ax, ay=ay,
                scale=scale, shift=shift, var=var, event_map=emap)


def make_jobs(n_reads, n_events, alphabet, k, table5, first_index=0, **kw):
    return [make_read(first_index + i, n_events, alphabet, k, table5, **kw) for i in range(n_reads)]


# ---- generating thousands of reads on several cores (bench.py, the full-size tests) -------------------------------------
_WORKER = {}


def _reads_chunk(task):
    """Worker (its own process, numpy only -- it never touches the GPU): reads `indices` of the workload `spec` names."""
    spec, indices = task
    key = (spec["kind"], spec["model"], spec.get("nhdp"))
    if _WORKER.get("key") != key:
        alpha, k, t10, tab = parse_model_table(spec["model"])
        _WORKER.clear()
        _WORKER.update(key=key, alpha=alpha, k=k, tab=tab)
        if spec["kind"] == "hdp":
            _WORKER["sampler"] = HdpSampler(parse_nhdp(spec["nhdp"]))
            _WORKER["pool"] = open(spec["ref_pool"]).read().split()[0].strip()
    w = _WORKER
    if spec["kind"] == "hdp":
        t5 = np.asarray(spec["table5"])
        return [make_read_hdp(int(i), spec["events"], w["alpha"], w["k"], t5, w["sampler"], w["pool"]) for i in indices]
    return [make_read(int(i), spec["events"], w["alpha"], w["k"], w["tab"], **spec.get("kw", {})) for i in indices]


def make_reads_parallel(spec, indices, workers=None):
    """make_read / make_read_hdp for every index, in `workers` spawned processes (default: up to 8, the CPU quota allowing);
    the result is identical to the serial loop (every read is seeded by its index)."""
    import os
    indices = [int(i) for i in indices]
    # SA_SYNTH_CACHE=<dir>: read sets are kept there (files of this code's own making) and loaded instead of generated again --
    # the counter passes of one profiling session run the same workload many times, and under a profiler the generator is serial
    cache = os.environ.get("SA_SYNTH_CACHE")
    cpath = None
    if cache and len(indices) >= 256:
        import hashlib
        import pickle
        key = repr((spec["kind"], os.path.basename(spec["model"]), os.path.basename(spec.get("nhdp") or ""), spec["events"],
                    sorted(spec.get("kw", {}).items()), indices[0], indices[-1], len(indices),
                    hashlib.sha1(np.asarray(spec["table5"]).tobytes()).hexdigest() if "table5" in spec else ""))
        cpath = os.path.join(cache, "reads_" + hashlib.sha1(key.encode()).hexdigest()[:20] + ".pkl")
        if os.path.exists(cpath):
            with open(cpath, "rb") as f:
                return pickle.load(f)

    def keep(reads):
        if cpath:
            os.makedirs(cache, exist_ok=True)
            with open(cpath + ".tmp%d" % os.getpid(), "wb") as f:
                pickle.dump(reads, f, protocol=4)
            os.replace(cpath + ".tmp%d" % os.getpid(), cpath)
        return reads
    if workers is None:
        workers = min(8, max(1, len(os.sched_getaffinity(0)) - 2))
        if os.environ.get("SA_SYNTH_WORKERS"):
            workers = int(os.environ["SA_SYNTH_WORKERS"])
        # under a profiler every child process would load the profiler's preloaded tool (and with it the GPU runtime): serial
        if any("rocprof" in v.lower() for v in (os.environ.get("LD_PRELOAD", ""), os.environ.get("ROCP_TOOL_LIBRARIES", ""),
                                                os.environ.get("ROCPROFILER_REGISTER_LIBRARY", ""))) or \
                any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
            workers = 1
    if workers <= 1 or len(indices) < 256:
        return keep(_reads_chunk((spec, indices)))
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    step = max(64, (len(indices) + 4 * workers - 1) // (4 * workers))
    tasks = [(spec, indices[a:a + step]) for a in range(0, len(indices), step)]
    try:
        with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as ex:
            parts = list(ex.map(_reads_chunk, tasks))
    except Exception:   # (no processes to be had: process limits, a broken pool) -- the serial loop gives the same reads
        return keep(_reads_chunk((spec, indices)))
    return keep([r for part in parts for r in part])
