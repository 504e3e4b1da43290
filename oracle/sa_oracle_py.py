"""ctypes binding of the CPU oracle (oracle/libsa_oracle.so) plus small text parsers.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under signalalign_amd/ may import this module.

The parsers here (model / npRead) are deliberately independent of the product's C loaders so the
two can be checked against each other.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EM_MEANONLY_DESCALED = 0
EM_TWODIST = 1
EM_TWODIST_DESCALED = 2
EM_HDP = 3


class Params(C.Structure):
    _fields_ = [("threshold", C.c_double), ("diagonal_expansion", C.c_int64),
                ("trace_back_diagonals", C.c_int64), ("min_diags_between_trace_back", C.c_int64),
                ("split_matrix_bigger_than_this", C.c_int64), ("constraint_diagonal_trim", C.c_int64)]


class Pair(C.Structure):
    _fields_ = [("prob_e7", C.c_int64), ("x", C.c_int64), ("y", C.c_int64), ("path", C.c_int32),
                ("kmer_id", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("cells_forward", C.c_double), ("cells_backward", C.c_double), ("n_tracebacks", C.c_int64),
                ("n_total_prob", C.c_int64), ("last_total_prob", C.c_double)]


class Job(C.Structure):
    _fields_ = [("ref", C.c_char_p), ("lX", C.c_int64), ("events", C.POINTER(C.c_double)), ("stride", C.c_int64),
                ("lY", C.c_int64), ("ax", C.POINTER(C.c_int64)), ("ay", C.POINTER(C.c_int64)),
                ("n_anchors", C.c_int64), ("scale", C.c_double), ("shift", C.c_double), ("var", C.c_double)]


PAIR_DTYPE = np.dtype([("prob_e7", "<i8"), ("x", "<i8"), ("y", "<i8"), ("path", "<i4"), ("kmer_id", "<i4")])


def build(force=False):
    so = os.path.join(_HERE, "libsa_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("sa_oracle.c", "sa_mea_oracle.c", "sa_hdp_oracle.c", "sa_oracle.h")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int64)
    L.sao_log_add.restype = C.c_double
    L.sao_log_add.argtypes = [C.c_double, C.c_double]
    L.sao_kmer_id.restype = C.c_int64
    L.sao_kmer_id.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    L.sao_diagonal_check.restype = C.c_int
    L.sao_diagonal_check.argtypes = [C.c_int64, C.c_int64, C.c_int64]
    L.sao_band.restype = C.c_int
    L.sao_band.argtypes = [ip, ip, C.c_int64, C.c_int64, C.c_int64, C.c_int64, ip, ip]
    L.sao_split_points.restype = C.c_int64
    L.sao_split_points.argtypes = [ip, ip, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, ip]
    L.sao_model_new.restype = C.c_void_p
    L.sao_model_new.argtypes = [C.c_char_p, C.c_int, C.c_int, dp, dp, C.c_int]
    L.sao_model_free.argtypes = [C.c_void_p]
    L.sao_model_set_read_params.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
    L.sao_model_scale_noise.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.sao_model_set_hdp.argtypes = [C.c_void_p, C.c_void_p]
    L.sao_model_set_to_hdp_expected_values.argtypes = [C.c_void_p]
    L.sao_model_match_table.restype = dp
    L.sao_model_match_table.argtypes = [C.c_void_p]
    L.sao_hdp_load.restype = C.c_void_p
    L.sao_hdp_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.sao_hdp_free.argtypes = [C.c_void_p]
    L.sao_hdp_density.restype = C.c_double
    L.sao_hdp_density.argtypes = [C.c_void_p, C.c_double, C.c_int64]
    L.sao_default_ambig.argtypes = [C.POINTER(C.c_char_p)]
    L.sao_expand_paths.restype = C.c_int64
    L.sao_expand_paths.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.c_int64]
    L.sao_align.restype = C.c_int64
    L.sao_align.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, dp, C.c_int64, C.c_int64, ip, ip, C.c_int64,
                            C.POINTER(Params), C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int,
                            C.POINTER(C.POINTER(Pair)), C.POINTER(Stats)]
    L.sao_expectations.restype = C.c_int64
    L.sao_expectations.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, dp, C.c_int64, C.c_int64, ip, ip, C.c_int64,
                                   C.POINTER(Params), C.POINTER(C.c_char_p), dp, dp, C.POINTER(ip),
                                   C.POINTER(dp), C.POINTER(Stats)]
    L.sao_expectations_ragged.restype = C.c_int64
    L.sao_expectations_ragged.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, dp, C.c_int64, C.c_int64, ip, ip, C.c_int64,
                                          C.POINTER(Params), C.POINTER(C.c_char_p), C.c_int, C.c_int, dp, dp, C.POINTER(ip),
                                          C.POINTER(dp), C.POINTER(Stats)]
    L.sao_kat_unbanded.restype = C.c_int64
    L.sao_kat_unbanded.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, dp, C.c_int64, C.c_int64, C.c_double,
                                   C.POINTER(C.c_char_p), dp, dp, dp, C.POINTER(C.POINTER(Pair))]
    L.sao_guide_to_anchors.restype = C.c_int64
    L.sao_guide_to_anchors.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int64, C.POINTER(C.c_int32), ip, C.c_int64,
                                       C.c_int64, ip, ip, C.c_int64]
    L.sao_filter_overlap.restype = C.c_int64
    L.sao_filter_overlap.argtypes = [ip, ip, C.c_int64, ip, ip]
    L.sao_remap_anchors.restype = C.c_int64
    L.sao_remap_anchors.argtypes = [ip, ip, C.c_int64, ip, C.c_int64, ip, ip]
    L.sao_estimate_params.restype = C.c_int
    L.sao_estimate_params.argtypes = [C.c_void_p, ip, dp, C.c_int64, C.c_char_p, C.c_int64, dp]
    L.sao_free.argtypes = [C.c_void_p]
    L.sao_align_batch_mt.restype = C.c_int
    L.sao_align_batch_mt.argtypes = [C.c_void_p, C.POINTER(Job), C.c_int64, C.POINTER(Params), C.c_int, ip, dp]
    L.sao_align_batch_mt2.restype = C.c_int
    L.sao_align_batch_mt2.argtypes = [C.c_void_p, C.POINTER(Job), C.c_int64, C.POINTER(Params), C.c_int, ip, dp, C.POINTER(C.c_char_p)]
    u8p = C.POINTER(C.c_uint8)
    L.sao_hdp_linspace.argtypes = [C.c_double, C.c_double, C.c_int64, dp]
    L.sao_hdp_spline_knot_slopes.argtypes = [dp, dp, C.c_int64, dp]
    L.sao_hdp_posterior_predictive.argtypes = [dp, dp, dp, C.c_int64]
    L.sao_hdp_prior_predictive.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, dp, dp, C.c_int64]
    L.sao_hdp_nig_posterior.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, dp, C.c_int64, dp]
    L.sao_hdp_distr_sample.restype = C.c_int
    L.sao_hdp_distr_sample.argtypes = [C.c_int64, ip, ip, ip, u8p, dp, C.c_int64, ip, ip, ip, dp, C.c_double, C.c_double, C.c_double,
                                       C.c_double, dp, C.c_int64, dp]
    _LIB = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def kmer_ids_of(model, seq, rna=False):
    """k-mer id of every position of a nucleotide string as build_kmer_list lists them (impl/eventAligner.c:772-790):
    for RNA, U reads as T and every k-mer is reversed."""
    k = model.k
    if rna:
        seq = seq.replace("U", "T")
        return np.array([model.kmer_id(seq[i:i + k][::-1]) for i in range(len(seq) - k + 1)], dtype=np.int32)
    return np.array([model.kmer_id(seq[i:i + k]) for i in range(len(seq) - k + 1)], dtype=np.int32)


def scalings_mom(model, event_means, kmer_ids):
    """estimate_scalings_using_mom (impl/eventAligner.c:784-843): returns (shift, scale)."""
    ev = np.ascontiguousarray(event_means, dtype=np.float64)
    ids = np.ascontiguousarray(kmer_ids, dtype=np.int32)
    sh, sc = C.c_double(), C.c_double()
    f = lib().sao_scalings_mom
    f.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_double),
                  C.POINTER(C.c_double)]
    f.restype = None
    f(model._h, _dp(ev), len(ev), ids.ctypes.data_as(C.POINTER(C.c_int32)), len(ids), C.byref(sh), C.byref(sc))
    return sh.value, sc.value


def event_align(model, event_means, kmer_ids):
    """adaptive_banded_simple_event_align (impl/eventAligner.c:899-1235).  Returns (kmer_idx, event_idx, status)."""
    ev = np.ascontiguousarray(event_means, dtype=np.float64)
    ids = np.ascontiguousarray(kmer_ids, dtype=np.int32)
    ko, eo, st = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.c_int()
    f = lib().sao_event_align
    f.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int32), C.c_int64,
                  C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int)]
    f.restype = C.c_int64
    n = f(model._h, _dp(ev), len(ev), ids.ctypes.data_as(C.POINTER(C.c_int32)), len(ids), C.byref(ko), C.byref(eo),
          C.byref(st))
    if n < 0:
        raise RuntimeError("sao_event_align failed: %d" % n)
    k = np.array([ko[i] for i in range(n)], dtype=np.int32)
    e = np.array([eo[i] for i in range(n)], dtype=np.int32)
    if n:
        lib().sao_free(ko)
        lib().sao_free(eo)
    return k, e, st.value


MEA_INF = 2 ** 31 - 1   # shortest_ref_per_event entry of an event without rows (the reference's np.inf)
MEA_STATUS = {0: "ok", 1: "empty", 2: "single event", 3: "no forward edge", 4: "no path", 5: "bad event index"}


def mea(rows, cols, data, shortest, return_all=False):
    """maximum_expected_accuracy_alignment + get_indexes_from_best_path (src/signalalign/mea_algorithm.py:25-264) on a
    COO matrix.  Returns (status, path [n, 2] of (ref, event), best sum) or, with return_all, the sums of all final
    forward edges as a fourth element."""
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    data = np.ascontiguousarray(data, dtype=np.float64)
    sh = np.asarray(shortest, dtype=np.float64)
    sh = np.ascontiguousarray(np.where(np.isfinite(sh), sh, MEA_INF).astype(np.int32))
    i32p = C.POINTER(C.c_int32)
    f = lib().sao_mea
    f.restype = C.c_int
    f.argtypes = [i32p, i32p, C.POINTER(C.c_double), C.c_int64, i32p, C.c_int64, C.POINTER(i32p), C.POINTER(i32p),
                  C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_int64)]
    pr, pe, ne, best = i32p(), i32p(), C.c_int64(), C.c_double()
    es, n_es = C.POINTER(C.c_double)(), C.c_int64()
    st = f(rows.ctypes.data_as(i32p), cols.ctypes.data_as(i32p), _dp(data), len(rows), sh.ctypes.data_as(i32p), len(sh),
           C.byref(pr), C.byref(pe), C.byref(ne), C.byref(best), C.byref(es), C.byref(n_es))
    path = np.zeros((ne.value, 2), dtype=np.int32)
    if ne.value:
        path[:, 0] = np.ctypeslib.as_array(pr, (ne.value,))
        path[:, 1] = np.ctypeslib.as_array(pe, (ne.value,))
    sums = np.ctypeslib.as_array(es, (n_es.value,)).copy() if n_es.value else np.zeros(0)
    for q in (pr, pe, es):
        if q:
            lib().sao_free(q)
    return (st, path, best.value, sums) if return_all else (st, path, best.value)


def mea_params(reference_index, event_index, posterior):
    """get_mea_params_from_events (mea_algorithm.py:267-320) in sparse form: (rows, cols, data, shortest_ref_per_event)."""
    ri = np.ascontiguousarray(reference_index, dtype=np.int64)
    ei = np.ascontiguousarray(event_index, dtype=np.int64)
    po = np.ascontiguousarray(posterior, dtype=np.float64)
    n = len(ri)
    rows = np.zeros(max(n, 1), dtype=np.int32)
    cols = np.zeros(max(n, 1), dtype=np.int32)
    data = np.zeros(max(n, 1), dtype=np.float64)
    n_ev = int(ei.max() - ei.min() + 1) if n else 1
    sh = np.zeros(n_ev, dtype=np.int32)
    i32p = C.POINTER(C.c_int32)
    f = lib().sao_mea_params
    f.restype = C.c_int64
    f.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_int64, i32p, i32p,
                  C.POINTER(C.c_double), i32p, C.POINTER(C.c_int64)]
    ne = C.c_int64()
    m = f(_ip(ri), _ip(ei), _dp(po), n, rows.ctypes.data_as(i32p), cols.ctypes.data_as(i32p), _dp(data),
          sh.ctypes.data_as(i32p), C.byref(ne))
    if m < 0:
        raise RuntimeError("sao_mea_params failed: %d" % m)
    return rows[:m].copy(), cols[:m].copy(), data[:m].copy(), sh


def mea_exhaustive(matrix, shortest):
    """Independent formulation used to cross-check sao_mea on small dense matrices, after the reference's `mea_slow`
    (mea_algorithm.py:726-816): every cell of an event looks at EVERY forward edge of the previous event.  Pure Python,
    small cases only.  Returns the best sum (0 when there is none)."""
    m = np.asarray(matrix, dtype=np.float64)
    n_ev, n_ref = m.shape
    front = []   # (ref, sum)
    started = False
    for ev in range(n_ev):
        top = 0.0
        new = []
        if not started:
            for r in range(n_ref):
                if 0 < m[ev, r] >= top:
                    new.append((r, m[ev, r]))
                    top = m[ev, r]
            if new:
                front, started = new, True
            continue
        opener = None
        for r in range(n_ref):   # a fresh start above every later reference position (:761-770)
            if m[ev, r] >= top and r < shortest[ev]:
                opener = (r, m[ev, r])
                top = m[ev, r]
        if opener is not None:
            new.append(opener)
        for r in range(n_ref):
            cand = [s + m[ev, r] for (fr, s) in front if fr < r] + [s for (fr, s) in front if fr == r]
            if cand:
                if max(cand) > top:
                    new.append((r, max(cand)))
                    top = max(cand)
            elif front[0][0] > r and m[ev, r] > top:
                new.append((r, m[ev, r]))
                top = m[ev, r]
        front = new
    return max([s for (_, s) in front], default=0.0)


def default_params(threshold=0.01, expansion=50, trace_back=100, min_diags=1000, split=3000 * 3000, trim=14):
    """signalMachine defaults (impl/signalMachine.c:487-490, :672-676) with Python's -g 100."""
    e = expansion if expansion % 2 == 0 else expansion + 1
    return Params(threshold, e, trace_back, min_diags, split, trim)


def ambig_map(table=None):
    """256-entry char* array; table=None -> create_ambig_bases() defaults; {} -> no ambiguity."""
    arr = (C.c_char_p * 256)()
    if table is None:
        lib().sao_default_ambig(arr)
    else:
        for k, v in table.items():
            arr[ord(k)] = v.encode()
    return arr


# ---------------------------------------------------------------------------------------------
# file parsers (independent of the product's C loaders)
# ---------------------------------------------------------------------------------------------
def parse_model_file(path):
    """impl/stateMachine.c:1440-1538: 3 whitespace-split lines."""
    with open(path) as f:
        l0 = f.readline().split()
        l1 = f.readline().split()
        l2 = f.readline().split()
    n_states, n_alpha, alphabet, k = int(l0[0]), int(l0[1]), l0[2], int(l0[3])
    t10 = np.array([float(t) for t in l1], dtype=np.float64)
    table = np.array([float(t) for t in l2], dtype=np.float64)
    assert n_states == 3 and len(t10) == 10 and len(table) == 5 * n_alpha ** k
    return dict(alphabet=alphabet, n_alpha=n_alpha, k=k, transitions10=t10, table5=table)


def parse_npread(path):
    """impl/nanopore.c:145-521: 14 whitespace-split lines."""
    with open(path) as f:
        lines = [f.readline() for _ in range(14)]
    h = lines[0].split()
    r = dict(read_length=int(h[0]), n_template_events=int(h[1]), n_complement_events=int(h[2]),
             template_read_length=int(h[3]), complement_read_length=int(h[4]))
    keys = ["scale", "shift", "var", "scale_sd", "var_sd", "drift"]
    r["template_params"] = {k: float(v) for k, v in zip(keys, h[5:11])}
    r["complement_params"] = {k: float(v) for k, v in zip(keys, h[11:17])}
    r["twoD"] = int(h[17])
    r["twoD_read"] = lines[1].strip()
    r["template_read"] = lines[2].strip()
    r["template_strand_event_map"] = np.array(lines[3].split(), dtype=np.int64)
    r["complement_read"] = lines[4].strip()
    r["complement_strand_event_map"] = np.array(lines[5].split(), dtype=np.int64)
    r["template_event_map"] = np.array(lines[6].split(), dtype=np.int64)
    r["template_events"] = np.array(lines[7].split(), dtype=np.float64).reshape(-1, 4)
    r["complement_event_map"] = np.array(lines[8].split(), dtype=np.int64)
    r["complement_events"] = np.array(lines[9].split(), dtype=np.float64).reshape(-1, 4)
    r["template_model_state"] = lines[10].split()
    r["template_p_model"] = np.array(lines[11].split(), dtype=np.float64)
    return r


# ---------------------------------------------------------------------------------------------
# object wrappers
# ---------------------------------------------------------------------------------------------
class Model:
    def __init__(self, alphabet, k, transitions10, table5, emission=EM_MEANONLY_DESCALED):
        self.alphabet = "".join(sorted(alphabet))
        self.k = k
        self.n_alpha = len(alphabet)
        t10 = np.ascontiguousarray(transitions10, dtype=np.float64)
        tb = np.ascontiguousarray(table5, dtype=np.float64)
        self._h = lib().sao_model_new(alphabet.encode(), len(alphabet), k, _dp(t10), _dp(tb), emission)
        self._hdp = None

    @classmethod
    def from_file(cls, path, emission=EM_MEANONLY_DESCALED):
        d = parse_model_file(path)
        return cls(d["alphabet"], d["k"], d["transitions10"], d["table5"], emission)

    def set_read_params(self, scale, shift, var):
        lib().sao_model_set_read_params(self._h, scale, shift, var)

    def scale_noise(self, scale_sd, var_sd):
        lib().sao_model_scale_noise(self._h, scale_sd, var_sd)

    def load_hdp(self, nhdp_path):
        alpha = C.create_string_buffer(64)
        na, k = C.c_int(), C.c_int()
        h = lib().sao_hdp_load(nhdp_path.encode(), alpha, C.byref(na), C.byref(k))
        if not h:
            raise IOError(nhdp_path)
        assert alpha.value.decode() == self.alphabet and k.value == self.k, (alpha.value, self.alphabet)
        self._hdp = h
        lib().sao_model_set_hdp(self._h, h)

    def set_to_hdp_expected_values(self):
        lib().sao_model_set_to_hdp_expected_values(self._h)

    def hdp_density(self, x, dp_id):
        return lib().sao_hdp_density(self._hdp, x, dp_id)

    def match_table(self):
        n = 5 * self.n_alpha ** self.k
        return np.ctypeslib.as_array(lib().sao_model_match_table(self._h), shape=(n,))

    def kmer_id(self, kmer):
        return lib().sao_kmer_id(kmer.encode(), self.alphabet.encode(), self.n_alpha, self.k)

    def expand_paths(self, kmer, ambig=None):
        amb = ambig if ambig is not None else ambig_map()
        out = (C.c_int32 * 65536)()
        n = lib().sao_expand_paths(self._h, kmer.encode(), amb, out, 65536)
        return n, list(out[:max(n, 0)])

    def __del__(self):
        try:
            if self._h:
                lib().sao_model_free(self._h)
                self._h = None
        except Exception:
            pass


def _events4(events):
    ev = np.ascontiguousarray(events, dtype=np.float64)
    if ev.ndim == 1:
        ev4 = np.zeros((len(ev), 4), dtype=np.float64)
        ev4[:, 0] = ev
        ev = ev4
    return ev


def align(model, ref, events, ax, ay, params=None, ambig=None, ragged=(1, 1), sort_output=True, want_stats=False):
    """Returns a structured array of pairs in the order signalMachine would write them."""
    p = params or default_params()
    ev = _events4(events)
    lX = max(len(ref) - (model.k - 1), 0)
    axa = np.ascontiguousarray(ax, dtype=np.int64)
    aya = np.ascontiguousarray(ay, dtype=np.int64)
    amb = ambig if ambig is not None else ambig_map()
    out = C.POINTER(Pair)()
    st = Stats()
    n = lib().sao_align(model._h, ref.encode(), lX, _dp(ev), ev.shape[1], ev.shape[0], _ip(axa), _ip(aya), len(axa),
                        C.byref(p), amb, ragged[0], ragged[1], 1 if sort_output else 0, C.byref(out), C.byref(st))
    if n < 0:
        raise RuntimeError("sao_align failed: %d" % n)
    res = np.zeros(n, dtype=PAIR_DTYPE)
    if n:
        C.memmove(res.ctypes.data, out, n * C.sizeof(Pair))
    lib().sao_free(out)
    return (res, st) if want_stats else res


def expectations(model, ref, events, ax, ay, params=None, ambig=None, ragged=(1, 1)):
    p = params or default_params()
    ev = _events4(events)
    lX = max(len(ref) - (model.k - 1), 0)
    axa = np.ascontiguousarray(ax, dtype=np.int64)
    aya = np.ascontiguousarray(ay, dtype=np.int64)
    amb = ambig if ambig is not None else ambig_map()
    trans = np.zeros(9, dtype=np.float64)
    lik = C.c_double(0.0)
    ak = C.POINTER(C.c_int64)()
    ae = C.POINTER(C.c_double)()
    st = Stats()
    n = lib().sao_expectations_ragged(model._h, ref.encode(), lX, _dp(ev), ev.shape[1], ev.shape[0], _ip(axa), _ip(aya),
                                      len(axa), C.byref(p), amb, int(ragged[0]), int(ragged[1]), _dp(trans), C.byref(lik),
                                      C.byref(ak), C.byref(ae), C.byref(st))
    if n < 0:
        raise RuntimeError("sao_expectations failed: %d" % n)
    pos = np.array([ak[i] for i in range(n)], dtype=np.int64)
    evs = np.array([ae[i] for i in range(n)], dtype=np.float64)
    lib().sao_free(ak)
    lib().sao_free(ae)
    return trans, lik.value, pos, evs, st


def kat_unbanded(model, ref, events, threshold, ambig=None):
    ev = _events4(events)
    lX = len(ref) - (model.k - 1)
    lY = ev.shape[0]
    amb = ambig if ambig is not None else ambig_map()
    tF, tB = C.c_double(), C.c_double()
    diag = np.zeros(lX + lY + 1, dtype=np.float64)
    out = C.POINTER(Pair)()
    n = lib().sao_kat_unbanded(model._h, ref.encode(), lX, _dp(ev), ev.shape[1], lY, threshold, amb, C.byref(tF),
                               C.byref(tB), _dp(diag), C.byref(out))
    if n < 0:
        raise RuntimeError("sao_kat_unbanded failed: %d" % n)
    res = np.zeros(n, dtype=PAIR_DTYPE)
    if n:
        C.memmove(res.ctypes.data, out, n * C.sizeof(Pair))
    lib().sao_free(out)
    return tF.value, tB.value, diag, res


def band(ax, ay, lX, lY, expansion):
    axa = np.ascontiguousarray(ax, dtype=np.int64)
    aya = np.ascontiguousarray(ay, dtype=np.int64)
    L = np.zeros(lX + lY + 1, dtype=np.int64)
    R = np.zeros(lX + lY + 1, dtype=np.int64)
    rc = lib().sao_band(_ip(axa), _ip(aya), len(axa), lX, lY, expansion, _ip(L), _ip(R))
    if rc != 0:
        raise ValueError("invalid diagonal")
    return L, R


def split_points(ax, ay, lX, lY, bigger, ragged_left, ragged_right):
    axa = np.ascontiguousarray(ax, dtype=np.int64)
    aya = np.ascontiguousarray(ay, dtype=np.int64)
    out = np.zeros(4 * (len(axa) + 2), dtype=np.int64)
    n = lib().sao_split_points(_ip(axa), _ip(aya), len(axa), lX, lY, bigger, int(ragged_left), int(ragged_right),
                               _ip(out))
    return out[:4 * n].reshape(-1, 4)


def guide_to_anchors(start1, end1, strand1, start2, ops, trim):
    """ops: list of (type, length) with type 0=M, 1=ref-only, 2=read-only."""
    t = np.array([o[0] for o in ops], dtype=np.int32)
    ln = np.array([o[1] for o in ops], dtype=np.int64)
    cap = int(ln.sum()) + 1
    ax = np.zeros(cap, dtype=np.int64)
    ay = np.zeros(cap, dtype=np.int64)
    n = lib().sao_guide_to_anchors(start1, end1, int(strand1), start2, t.ctypes.data_as(C.POINTER(C.c_int32)), _ip(ln),
                                   len(t), trim, _ip(ax), _ip(ay), cap)
    return ax[:n].copy(), ay[:n].copy()


def remap_anchors(ax, ay, event_map, map_offset):
    axa = np.ascontiguousarray(ax, dtype=np.int64)
    aya = np.ascontiguousarray(ay, dtype=np.int64)
    em = np.ascontiguousarray(event_map, dtype=np.int64)
    ox = np.zeros(len(axa) + 1, dtype=np.int64)
    oy = np.zeros(len(axa) + 1, dtype=np.int64)
    n = lib().sao_remap_anchors(_ip(axa), _ip(aya), len(axa), _ip(em), map_offset, _ip(ox), _ip(oy))
    return ox[:n].copy(), oy[:n].copy()


def estimate_params(model, strand_event_map, events4, strand_read):
    """Modifies events4 in place (drift) and the model's noise columns; returns dict of 7 params."""
    em = np.ascontiguousarray(strand_event_map, dtype=np.int64)
    assert events4.flags["C_CONTIGUOUS"] and events4.shape[1] == 4
    out = np.zeros(7, dtype=np.float64)
    rc = lib().sao_estimate_params(model._h, _ip(em), _dp(events4), events4.shape[0], strand_read.encode(),
                                   len(strand_read), _dp(out))
    if rc != 0:
        raise RuntimeError("sao_estimate_params failed: %d" % rc)
    return dict(zip(["scale", "shift", "var", "drift", "scale_sd", "var_sd", "shift_sd"], out.tolist()))


def align_batch_mt(model, jobs, params, n_threads, ambig=None):
    """jobs: list of dicts(ref, events4, ax, ay, scale, shift, var). Returns (n_pairs, cells) arrays.  ambig: an ambig_map()
    (None: create_ambig_bases' defaults)."""
    n = len(jobs)
    arr = (Job * n)()
    keep = []
    for i, j in enumerate(jobs):
        ev = _events4(j["events"])
        axa = np.ascontiguousarray(j["ax"], dtype=np.int64)
        aya = np.ascontiguousarray(j["ay"], dtype=np.int64)
        rb = j["ref"].encode()
        keep.append((ev, axa, aya, rb))
        arr[i] = Job(rb, len(j["ref"]) - (model.k - 1), _dp(ev), ev.shape[1], ev.shape[0], _ip(axa), _ip(aya),
                     len(axa), j["scale"], j["shift"], j["var"])
    npairs = np.zeros(n, dtype=np.int64)
    cells = np.zeros(n, dtype=np.float64)
    lib().sao_align_batch_mt2(model._h, arr, n, C.byref(params), n_threads, _ip(npairs), _dp(cells), ambig)
    return npairs, cells


# ---- HDP rebuild, the deterministic pieces (sa_hdp_oracle.c) ----------------------------------------------------------
def hdp_linspace(start, stop, length):
    out = np.zeros(int(length), dtype=np.float64)
    lib().sao_hdp_linspace(float(start), float(stop), int(length), _dp(out))
    return out


def hdp_spline_knot_slopes(x, y):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    k = np.zeros(len(x), dtype=np.float64)
    lib().sao_hdp_spline_knot_slopes(_dp(x), _dp(y), len(x), _dp(k))
    return k


def hdp_posterior_predictive(params5, x):
    p5 = np.ascontiguousarray(params5, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros(len(x), dtype=np.float64)
    lib().sao_hdp_posterior_predictive(_dp(p5), _dp(x), _dp(out), len(x))
    return out


def hdp_prior_predictive(mu, nu, two_alpha, beta, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros(len(x), dtype=np.float64)
    lib().sao_hdp_prior_predictive(float(mu), float(nu), float(two_alpha), float(beta), _dp(x), _dp(out), len(x))
    return out


def hdp_nig_posterior(mu, nu, two_alpha, beta, data):
    d = np.ascontiguousarray(data, dtype=np.float64)
    out = np.zeros(5, dtype=np.float64)
    lib().sao_hdp_nig_posterior(float(mu), float(nu), float(two_alpha), float(beta), _dp(d), len(d), _dp(out))
    return out


def hdp_distr_sample(dp_parent, dp_nfc, dp_depth, observed, gamma, f_type, f_parent, f_dp, f_params, mu, nu, two_alpha, beta, grid):
    """take_distr_sample from arrays: returns the collectors, num_dps x grid_length (rows of unobserved DPs are zero)"""
    a = [np.ascontiguousarray(v, dtype=np.int64) for v in (dp_parent, dp_nfc, dp_depth)]
    ob = np.ascontiguousarray(observed, dtype=np.uint8)
    g = np.ascontiguousarray(gamma, dtype=np.float64)
    f = [np.ascontiguousarray(v, dtype=np.int64) for v in (f_type, f_parent, f_dp)]
    fp = np.ascontiguousarray(f_params, dtype=np.float64)
    grid = np.ascontiguousarray(grid, dtype=np.float64)
    out = np.zeros((len(a[0]), len(grid)), dtype=np.float64)
    rc = lib().sao_hdp_distr_sample(len(a[0]), _ip(a[0]), _ip(a[1]), _ip(a[2]), ob.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(g), len(f[0]),
                                    _ip(f[0]), _ip(f[1]), _ip(f[2]), _dp(fp), float(mu), float(nu), float(two_alpha), float(beta),
                                    _dp(grid), len(grid), _dp(out))
    if rc != 0:
        raise RuntimeError("sao_hdp_distr_sample failed")
    return out
