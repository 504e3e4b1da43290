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


def make_read(index, n_events, alphabet, k, table5, trim=14, cpg_ambiguous=False, thin_anchors=0.0):
    """Returns dict(ref, events4, ax, ay, scale, shift, var, event_map, read) for read `index`.

    cpg_ambiguous: replace every C that is followed by G with 'X' (config 3: CpG cytosines ambiguous).
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
    if cpg_ambiguous:
        ref = read.replace("CG", "XG")
    return dict(ref=ref, read=read, events4=events4, events=np.ascontiguousarray(events4[lo:hi]), ax=ax, ay=ay,
                scale=scale, shift=shift, var=var, event_map=emap)


def make_jobs(n_reads, n_events, alphabet, k, table5, first_index=0, **kw):
    return [make_read(first_index + i, n_events, alphabet, k, table5, **kw) for i in range(n_reads)]
