"""ctypes mirror of include/signalalign_hip.h.  No arithmetic happens here."""
import ctypes as C
import time
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FLAG_EXACT = 1
FLAG_FORCE_GENERIC = 2
FLAG_DEVICE_TO_ITSELF = 8
FLAG_INPUTS_IN_HOST_BLOCK = 16
FLAG_VC_ROWS = 32
FLAG_PAIRS8 = 64


class SaError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = lib().sa_strerror(code).decode() if _LIB is not None else str(code)
        super().__init__("%s: %s (%d)" % (what, msg, code))


class Params(C.Structure):
    _fields_ = [("threshold", C.c_double), ("diagonal_expansion", C.c_int64), ("trace_back_diagonals", C.c_int64),
                ("min_diags_between_trace_back", C.c_int64), ("split_matrix_bigger_than_this", C.c_int64)]


class HdpDesc(C.Structure):
    _fields_ = [("num_dps", C.c_int64), ("grid_length", C.c_int64), ("grid_start", C.c_double),
                ("grid_stop", C.c_double), ("parent", C.POINTER(C.c_int64)), ("observed", C.POINTER(C.c_uint8)),
                ("post_pred", C.POINTER(C.POINTER(C.c_double))), ("slopes", C.POINTER(C.POINTER(C.c_double)))]


class Job(C.Structure):
    _fields_ = [("ref", C.c_char_p), ("ref_len", C.c_int64), ("events", C.POINTER(C.c_double)),
                ("event_stride", C.c_int64), ("n_events", C.c_int64), ("anchor_x", C.POINTER(C.c_int64)),
                ("anchor_y", C.POINTER(C.c_int64)), ("n_anchors", C.c_int64), ("scale", C.c_double),
                ("shift", C.c_double), ("var", C.c_double), ("ends", C.c_uint)]


JOB_LEFT_END_NOT_RAGGED, JOB_RIGHT_END_NOT_RAGGED = 1, 2   # sa_job_t.ends (alignmentHasRaggedLeftEnd / RightEnd, inverted)


class Pair(C.Structure):
    _fields_ = [("prob_e7", C.c_int64), ("x", C.c_int32), ("y", C.c_int32), ("path", C.c_int32),
                ("kmer_id", C.c_int32)]


class BatchStats(C.Structure):
    _fields_ = [("cells_forward", C.c_double), ("cells_backward", C.c_double), ("ms_forward", C.c_double),
                ("ms_backward", C.c_double), ("ms_fold", C.c_double), ("ms_total_device", C.c_double),
                ("f_bytes", C.c_double), ("n_regions", C.c_int64), ("n_segments", C.c_int64),
                ("n_checkpoints", C.c_int64), ("n_fast_regions", C.c_int64), ("n_chunks", C.c_int64),
                ("n_groups", C.c_int64), ("n_ring_regions", C.c_int64), ("n_strip_regions", C.c_int64),
                ("device_bytes", C.c_double)]


class EaJob(C.Structure):
    _fields_ = [("sequence", C.c_char_p), ("seq_len", C.c_int64), ("event_mean", C.POINTER(C.c_double)),
                ("n_events", C.c_int64), ("scale", C.c_double), ("shift", C.c_double), ("var", C.c_double)]


class MeaJob(C.Structure):
    _fields_ = [("event_idx", C.POINTER(C.c_int32)), ("ref_idx", C.POINTER(C.c_int32)), ("posterior", C.POINTER(C.c_double)),
                ("n", C.c_int64), ("shortest_ref_per_event", C.POINTER(C.c_int32)), ("n_events", C.c_int64)]


MEA_INF = 2 ** 31 - 1
MEA_STATUS = {0: "ok", 1: "empty", 2: "single event", 3: "no forward edge", 4: "no path", 5: "bad event index"}


class PlanInfo(C.Structure):
    _fields_ = [("n_regions", C.c_int64), ("n_segments", C.c_int64), ("n_checkpoints", C.c_int64),
                ("cells_forward", C.c_double), ("cells_backward", C.c_double), ("f_cellpaths", C.c_int64),
                ("max_span", C.c_int64), ("n_fast_regions", C.c_int64), ("n_ring_regions", C.c_int64)]


PAIR_DTYPE = np.dtype([("prob_e7", "<i8"), ("x", "<i4"), ("y", "<i4"), ("path", "<i4"), ("kmer_id", "<i4")])

EXPORTS = ["sa_model_create", "sa_model_load", "sa_model_destroy", "sa_model_alphabet", "sa_model_table5",
           "sa_model_set_to_hdp_expected_values", "sa_model_set_emission", "sa_model_clone_with_table", "sa_kmer_id", "sa_default_ambig", "sa_load_ambig",
           "sa_batch_create", "sa_batch_create_deferred", "sa_batch_prepare", "sa_batch_run", "sa_batch_n_pairs", "sa_batch_all_pairs_summary", "sa_batch_pairs", "sa_batch_pairs16", "sa_batch_pairs16_all", "sa_batch_pairs8", "sa_batch_pairs8_all", "sa_batch_pairs_all", "sa_batch_stats",
           "sa_batch_job_cells", "sa_batch_release_device", "sa_batch_destroy", "sa_align_batch", "sa_expect_batch", "sa_expect_last_stats", "sa_plan_describe", "sa_plan_digest",
           "sa_plan_check_path_records", "sa_dplan_compare",
           "sa_guide_to_anchors", "sa_remap_anchors", "sa_estimate_params", "sa_scalings_mom", "sa_event_align_batch", "sa_event_align_release", "sa_pool_release", "sa_pool_release_device", "sa_pool_configure", "sa_host_alloc", "sa_host_free", "sa_pair_roundtrip", "sa_fasta_subsequence", "sa_format_f6", "sa_batch_start", "sa_batch_wait", "sa_mea_batch", "sa_mea_release", "sa_mea_params", "sa_batch_mea", "sa_mea_printed_posterior", "sa_mea_printed_posterior_device", "sa_device_count", "sa_device_memory", "sa_strerror", "sa_hdp_state_load", "sa_hdp_state_write", "sa_hdp_state_info", "sa_hdp_state_free", "sa_hdp_state_distr_sample", "sa_hdp_state_sample_weights", "sa_hdp_finalize_distributions",
           "sa_hdp_state_new", "sa_hdp_state_new_tree", "sa_hdp_nig_params_from_table", "sa_hdp_state_pass_data", "sa_hdp_state_pass_assignments", "sa_hdp_state_pass_assignment_file", "sa_hdp_state_kmer_dp", "sa_hdp_state_gibbs", "sa_hdp_state_finalize", "sa_hdp_state_samples_taken", "sa_hdp_digamma", "sa_hdp_trigamma",
           "sa_hmm_create", "sa_hmm_destroy", "sa_hmm_view", "sa_hmm_set_event_model", "sa_hmm_add_expectations",
           "sa_hmm_add_emission_expectation", "sa_hmm_add_assignment", "sa_hmm_add_expectations_file", "sa_hmm_write", "sa_hmm_load", "sa_hmm_normalize",
           "sa_hmm_load_into_model", "sa_model_transitions10",
           "sa_version", "sa_free"]


class HmmView(C.Structure):
    """sa_hmm_view_t (include/signalalign_hip.h)"""
    _fields_ = [("type", C.c_int), ("n_states", C.c_int), ("n_alpha", C.c_int), ("k", C.c_int), ("alphabet", C.c_char * 64),
                ("n_kmers", C.c_int64), ("transitions", C.POINTER(C.c_double)), ("likelihood", C.POINTER(C.c_double)),
                ("event_model", C.POINTER(C.c_double)), ("event_expectations", C.POINTER(C.c_double)),
                ("posteriors", C.POINTER(C.c_double)), ("observed", C.POINTER(C.c_uint8)), ("threshold", C.c_double),
                ("n_assignments", C.c_int64), ("assignment_events", C.POINTER(C.c_double)),
                ("assignment_kmers", C.POINTER(C.c_char)), ("has_model", C.c_int)]


class HdpStateInfo(C.Structure):
    """sa_hdp_state_info_t (include/signalalign_hip.h)"""
    _fields_ = ([(n, C.c_int64) for n in ("num_dps", "depth", "grid_length", "n_data", "n_factors", "n_base_factors", "n_observed",
                                          "base_dp", "alphabet_size", "kmer_length")] +
                [(n, C.c_double) for n in ("mu", "nu", "alpha", "beta", "grid_start", "grid_stop")] +
                [(n, C.c_int) for n in ("splines_finalized", "has_data", "sample_gamma")] +
                [("data", C.POINTER(C.c_double)), ("data_dp", C.POINTER(C.c_int64)), ("gamma", C.POINTER(C.c_double)),
                 ("grid", C.POINTER(C.c_double)), ("dp_parent", C.POINTER(C.c_int64)),
                 ("dp_num_factor_children", C.POINTER(C.c_int64)), ("dp_depth", C.POINTER(C.c_int64)),
                 ("observed", C.POINTER(C.c_uint8)), ("row_of_dp", C.POINTER(C.c_int64)), ("post", C.POINTER(C.c_double)),
                 ("slope", C.POINTER(C.c_double)), ("f_type", C.POINTER(C.c_int64)), ("f_parent", C.POINTER(C.c_int64)),
                 ("f_ref", C.POINTER(C.c_int64)), ("f_params", C.POINTER(C.c_double)), ("f_n_children", C.POINTER(C.c_int64))])


def library_path():
    # SA_LIBRARY: load another build of the same ABI (probes/host_asan.sh: the host sources under AddressSanitizer)
    return os.environ.get("SA_LIBRARY") or os.path.join(_HERE, "lib", "libsignalalign_hip.so")


def build(force=False):
    """Compile the HIP library and the CLI in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    so = library_path()
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    srcs.append(os.path.join(_HERE, "..", "include", "signalalign_hip.h"))
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale and not os.environ.get("SA_LIBRARY"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = library_path()
    if not os.path.exists(so):
        raise ImportError("libsignalalign_hip.so is not built (run __graft_entry__.build()); there is no fallback")
    L = C.CDLL(so)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
    L.sa_strerror.restype = C.c_char_p
    L.sa_strerror.argtypes = [C.c_int]
    L.sa_version.restype = C.c_char_p
    L.sa_device_count.restype = C.c_int
    L.sa_model_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_int, dp, dp, C.POINTER(HdpDesc)]
    L.sa_model_load.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p]
    L.sa_model_destroy.argtypes = [C.c_void_p]
    L.sa_model_alphabet.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.sa_model_table5.restype = dp
    L.sa_model_table5.argtypes = [C.c_void_p]
    L.sa_model_set_to_hdp_expected_values.argtypes = [C.c_void_p]
    L.sa_model_set_emission.argtypes = [C.c_void_p, C.c_int]
    L.sa_model_transitions10.argtypes = [C.c_void_p, dp]
    L.sa_hmm_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
    L.sa_hmm_destroy.argtypes = [C.c_void_p]
    L.sa_hmm_destroy.restype = None
    L.sa_hmm_view.argtypes = [C.c_void_p, C.POINTER(HmmView)]
    L.sa_hmm_set_event_model.argtypes = [C.c_void_p, dp]
    L.sa_hmm_add_expectations.argtypes = [C.c_void_p, dp, C.c_double]
    L.sa_hmm_add_emission_expectation.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_double]
    L.sa_hmm_add_assignment.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    L.sa_hmm_write.argtypes = [C.c_void_p, C.c_char_p]
    L.sa_hmm_load.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.c_double, C.c_double]
    L.sa_hmm_normalize.argtypes = [C.c_void_p]
    L.sa_hmm_add_expectations_file.argtypes = [C.c_void_p, C.c_char_p]
    L.sa_hmm_load_into_model.argtypes = [C.c_void_p, C.c_void_p]
    L.sa_kmer_id.restype = C.c_int64
    L.sa_kmer_id.argtypes = [C.c_void_p, C.c_char_p]
    L.sa_default_ambig.argtypes = [C.POINTER(C.c_char_p)]
    L.sa_load_ambig.argtypes = [C.c_char_p, C.POINTER(C.c_char_p)]
    L.sa_batch_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(Params), C.POINTER(Job), C.c_int64,
                                  C.POINTER(C.c_char_p), C.c_int, C.c_uint]
    L.sa_batch_create_deferred.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(Params), C.POINTER(Job), C.c_int64,
                                  C.POINTER(C.c_char_p), C.c_int, C.c_uint]
    L.sa_batch_run.argtypes = [C.c_void_p]
    L.sa_plan_check_path_records.restype = C.c_int64
    L.sa_plan_check_path_records.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Job), C.POINTER(C.c_char_p), ip]
    L.sa_dplan_compare.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Job), C.c_int64, C.POINTER(C.c_char_p), C.c_int,
                                   C.c_uint]
    L.sa_pool_configure.argtypes = [C.c_int64, C.c_int64]
    L.sa_host_alloc.argtypes = [C.c_size_t]
    L.sa_host_alloc.restype = C.c_void_p
    L.sa_host_free.argtypes = [C.c_void_p]
    L.sa_host_free.restype = None
    L.sa_batch_n_pairs.argtypes = [C.c_void_p, C.c_int64, ip]
    L.sa_batch_all_pairs_summary.argtypes = [C.c_void_p, C.c_int64, ip, ip]
    L.sa_batch_pairs.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
    L.sa_batch_pairs16.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), ip]
    L.sa_batch_pairs_all.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, ip]
    L.sa_batch_pairs16_all.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), ip]
    L.sa_batch_pairs8.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), ip]
    L.sa_batch_pairs8_all.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), ip]
    L.sa_batch_stats.argtypes = [C.c_void_p, C.POINTER(BatchStats)]
    L.sa_batch_job_cells.argtypes = [C.c_void_p, C.c_int64, dp, dp]
    L.sa_batch_destroy.argtypes = [C.c_void_p]
    L.sa_fasta_subsequence.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
    L.sa_batch_start.argtypes = [C.c_void_p]
    L.sa_batch_prepare.argtypes = [C.c_void_p]
    L.sa_batch_wait.argtypes = [C.c_void_p]
    L.sa_plan_describe.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Job), C.POINTER(C.c_char_p), C.c_uint,
                                   C.POINTER(PlanInfo), ip, C.c_int64, ip, C.c_int64, ip, C.c_int64]
    L.sa_plan_digest.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Job), C.c_int64, C.POINTER(C.c_char_p), C.c_uint,
                                 C.c_int, C.POINTER(PlanInfo), C.POINTER(C.c_uint64)]
    L.sa_guide_to_anchors.restype = C.c_int64
    L.sa_guide_to_anchors.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int64, C.POINTER(C.c_int32), ip, C.c_int64,
                                      C.c_int64, ip, ip, C.c_int64]
    L.sa_remap_anchors.restype = C.c_int64
    L.sa_remap_anchors.argtypes = [ip, ip, C.c_int64, ip, C.c_int64, ip, ip]
    L.sa_estimate_params.argtypes = [C.c_void_p, dp, ip, dp, C.c_int64, C.c_char_p, C.c_int64, dp]
    L.sa_expect_batch.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Job), C.c_int64, C.POINTER(C.c_char_p),
                                  C.c_int, C.c_uint, dp, dp, C.POINTER(C.c_void_p), ip]
    L.sa_expect_last_stats.argtypes = [C.POINTER(BatchStats)]
    L.sa_scalings_mom.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, dp, C.c_int64, C.c_uint, dp, dp]
    L.sa_event_align_batch.argtypes = [C.c_void_p, C.POINTER(EaJob), C.c_int64, C.c_int, C.c_uint, C.POINTER(C.c_void_p), ip,
                                       C.POINTER(C.c_int32), dp, dp]
    i32p = C.POINTER(C.c_int32)
    L.sa_mea_batch.argtypes = [C.POINTER(MeaJob), C.c_int64, C.c_int, C.c_uint, C.POINTER(C.c_void_p), ip, dp, i32p, i32p, dp]
    L.sa_batch_mea.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_void_p), ip, dp, i32p, dp]
    L.sa_mea_printed_posterior_device.argtypes = [C.c_int64, C.c_int64, dp, C.c_int]
    L.sa_mea_printed_posterior.restype = C.c_double
    L.sa_mea_printed_posterior.argtypes = [C.c_int64]
    L.sa_mea_params.restype = C.c_int64
    L.sa_mea_params.argtypes = [ip, ip, dp, C.c_int64, i32p, i32p, dp, i32p, ip]
    L.sa_free.argtypes = [C.c_void_p]
    L.sa_hdp_state_load.argtypes = [C.POINTER(C.c_void_p), C.c_char_p]
    L.sa_hdp_state_write.argtypes = [C.c_void_p, C.c_char_p]
    L.sa_hdp_state_info.argtypes = [C.c_void_p, C.POINTER(HdpStateInfo)]
    L.sa_hdp_state_free.argtypes = [C.c_void_p]
    L.sa_hdp_state_distr_sample.argtypes = [C.c_void_p, C.c_int, dp]
    L.sa_hdp_state_sample_weights.argtypes = [C.c_void_p, C.POINTER(ip), C.POINTER(ip), C.POINTER(dp), ip]
    L.sa_hdp_finalize_distributions.argtypes = [dp, C.c_int64, dp, C.c_int64, C.c_int64, C.c_int, dp, dp]
    L.sa_hdp_state_new.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_int64, ip, dp, dp, dp, C.c_double, C.c_double, C.c_int64,
                                   C.c_double, C.c_double, C.c_double, C.c_double]
    L.sa_hdp_state_new_tree.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int64, ip, dp, dp, dp, C.c_double, C.c_double, C.c_int64,
                                        C.c_double, C.c_double, C.c_double, C.c_double]
    L.sa_hdp_nig_params_from_table.argtypes = [dp, C.c_int64, dp, dp, dp, dp]
    L.sa_hdp_state_pass_data.argtypes = [C.c_void_p, dp, ip, C.c_int64]
    L.sa_hdp_state_pass_assignments.argtypes = [C.c_void_p, C.c_char_p, dp, C.c_int64]
    L.sa_hdp_state_pass_assignment_file.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, ip]
    L.sa_hdp_state_kmer_dp.argtypes = [C.c_void_p, C.c_char_p]
    L.sa_hdp_state_gibbs.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_int, C.c_int]
    L.sa_hdp_state_finalize.argtypes = [C.c_void_p, C.c_int]
    L.sa_hdp_state_samples_taken.argtypes = [C.c_void_p]
    L.sa_hdp_state_samples_taken.restype = C.c_int64
    L.sa_hdp_digamma.argtypes = [C.c_double]
    L.sa_hdp_digamma.restype = C.c_double
    L.sa_hdp_trigamma.argtypes = [C.c_double]
    L.sa_hdp_trigamma.restype = C.c_double
    _LIB = L
    return L


def _chk(rc, what):
    if rc != 0:
        raise SaError(rc, what)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def device_count():
    return lib().sa_device_count()


def default_params(threshold=0.01, expansion=50, trace_back=100, min_diags=1000, split=3000 * 3000):
    """signalMachine defaults (impl/signalMachine.c:487-490) with the Python driver's -g 100."""
    e = expansion if expansion % 2 == 0 else expansion + 1
    return Params(threshold, e, trace_back, min_diags, split)


def default_ambig(table=None):
    arr = (C.c_char_p * 256)()
    if table is None:
        lib().sa_default_ambig(arr)
    else:
        for k, v in table.items():
            arr[ord(k)] = v.encode()
    return arr


class Model:
    """StateMachine3 / StateMachine3_HDP as the C library holds it."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def create(cls, alphabet, k, transitions10, table5, hdp=None):
        h = C.c_void_p()
        t10 = np.ascontiguousarray(transitions10, dtype=np.float64)
        tb = np.ascontiguousarray(table5, dtype=np.float64)
        _chk(lib().sa_model_create(C.byref(h), 3, alphabet.encode(), k, _dp(t10), _dp(tb), hdp), "sa_model_create")
        return cls(h)

    @classmethod
    def load(cls, model_path, nhdp_path=None):
        h = C.c_void_p()
        _chk(lib().sa_model_load(C.byref(h), model_path.encode(), nhdp_path.encode() if nhdp_path else None),
             "sa_model_load")
        return cls(h)

    def alphabet(self):
        buf = C.create_string_buffer(64)
        na, k = C.c_int(), C.c_int()
        lib().sa_model_alphabet(self._h, buf, C.byref(na), C.byref(k))
        return buf.value.decode(), k.value

    def table5(self):
        alpha, k = self.alphabet()
        n = 5 * len(alpha) ** k
        return np.ctypeslib.as_array(lib().sa_model_table5(self._h), shape=(n,))

    def kmer_id(self, kmer):
        return lib().sa_kmer_id(self._h, kmer.encode())

    def transitions10(self):
        """the model's transitions as the ten tokens of a .model file's second line (linear space)"""
        out = np.zeros(10, dtype=np.float64)
        _chk(lib().sa_model_transitions10(self._h, _dp(out)), "sa_model_transitions10")
        return out

    def set_emission(self, emission):
        """0: MeanOnly (signalMachine's), 1: the two-distribution emission (sa_model_set_emission)."""
        _chk(lib().sa_model_set_emission(self._h, int(emission)), "sa_model_set_emission")

    def set_to_hdp_expected_values(self):
        _chk(lib().sa_model_set_to_hdp_expected_values(self._h), "sa_model_set_to_hdp_expected_values")

    def close(self):
        if self._h:
            lib().sa_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _make_jobs(jobs):
    """jobs: list of dicts(ref:str, events: (n,) or (n,4) float64, ax, ay, scale, shift, var[, ragged=(left, right)]);
    ragged: the last two arguments of getAlignedPairsUsingAnchors, default (1, 1) as signalMachine passes them."""
    n = len(jobs)
    arr = (Job * max(n, 1))()
    keep = []
    for i, j in enumerate(jobs):
        ev = np.ascontiguousarray(j["events"], dtype=np.float64)
        stride = 1 if ev.ndim == 1 else ev.shape[1]
        ax = np.ascontiguousarray(j["ax"], dtype=np.int64)
        ay = np.ascontiguousarray(j["ay"], dtype=np.int64)
        rb = j["ref"].encode() if isinstance(j["ref"], str) else j["ref"]
        keep.append((ev, ax, ay, rb))
        rl, rr = j.get("ragged", (1, 1))
        arr[i] = Job(rb, len(rb), _dp(ev), stride, ev.shape[0], _ip(ax), _ip(ay), len(ax), j.get("scale", 1.0),
                     j.get("shift", 0.0), j.get("var", 1.0),
                     (0 if rl else JOB_LEFT_END_NOT_RAGGED) | (0 if rr else JOB_RIGHT_END_NOT_RAGGED))
    return arr, keep


class HostBlock:
    """sa_host_alloc: one page-locked block; empty(shape, dtype) carves 8-byte aligned numpy arrays out of it (for the event
    records and anchor arrays of jobs passed with FLAG_INPUTS_IN_HOST_BLOCK)."""

    def __init__(self, nbytes):
        L = lib()
        self.nbytes = int(nbytes)
        self._p = L.sa_host_alloc(self.nbytes)
        if not self._p:
            raise SaError(-2, "sa_host_alloc")
        self._buf = (C.c_char * self.nbytes).from_address(self._p)
        self._used = 0

    def empty(self, shape, dtype):
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
        off = (self._used + 7) & ~7
        if off + n * dt.itemsize > self.nbytes:
            raise ValueError("HostBlock: full")
        self._used = off + n * dt.itemsize
        return np.frombuffer(self._buf, dtype=dt, count=n, offset=off).reshape(shape)

    def close(self):
        if self._p:
            self._buf = None
            lib().sa_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def jobs_bytes_in_block(jobs):
    """Bytes a HostBlock needs for these jobs' event records and anchors."""
    n = 0
    for j in jobs:
        n += (np.asarray(j["events"]).size * 8 + 7) & ~7
        n += 2 * 8 * len(j["ax"])
    return n + 64


class JobArray:
    """A list of jobs marshalled once into the C array sa_batch_create takes (a C caller has it anyway; building it costs
    Python several milliseconds per thousand reads).  Pass it to Batch in place of the list.
    host_block=True: the event records and anchor arrays are copied into one sa_host_alloc block (self.block), as a caller
    that reads its inputs straight into page-locked memory has them; pass FLAG_INPUTS_IN_HOST_BLOCK with it."""

    def __init__(self, jobs, host_block=False, interleaved=False):
        self.n = len(jobs)
        self.block = None
        if host_block:
            self.block = HostBlock(jobs_bytes_in_block(jobs))
            moved = [dict(j) for j in jobs]
            if interleaved:          # read after read: events, anchors, events, ...
                order = [(q, key) for q in moved for key in ("events", "ax", "ay")]
            else:                    # all event records, then all anchors (the library then sends the block in two pieces)
                order = [(q, "events") for q in moved] + [(q, key) for q in moved for key in ("ax", "ay")]
            for q, key in order:
                src = np.asarray(q[key], dtype=np.float64 if key == "events" else np.int64)
                q[key] = self.block.empty(src.shape, src.dtype)
                q[key][...] = src
            jobs = moved
        self.jobs = jobs    # (host_block: the dicts whose arrays live in the block)
        self.arr, self._keep = _make_jobs(jobs)


class Batch:
    """sa_batch_t: plan + HBM-resident inputs; run() launches the kernels."""

    def __init__(self, model, params, jobs, ambig=None, device=0, flags=0, deferred=False):
        """deferred=True: sa_batch_create_deferred (the plan is collected on the batch's first use; the job arrays and the
        ambiguity table are kept alive by this object)."""
        self._h = C.c_void_p()
        if isinstance(jobs, JobArray):
            self.n_jobs, arr, self._keep = jobs.n, jobs.arr, jobs
        else:
            self.n_jobs = len(jobs)
            arr, self._keep = _make_jobs(jobs)
        amb = ambig if ambig is not None else default_ambig()
        self._amb, self._model = amb, model
        self._flags = int(flags)
        fn, name = (lib().sa_batch_create_deferred, "sa_batch_create_deferred") if deferred else (lib().sa_batch_create, "sa_batch_create")
        _chk(fn(C.byref(self._h), model._h, C.byref(params), arr, self.n_jobs, amb, device, flags), name)

    def run(self):
        _chk(lib().sa_batch_run(self._h), "sa_batch_run")

    def release_device(self):
        """sa_batch_release_device: the working storage in HBM back to the allocator, the results kept"""
        _chk(lib().sa_batch_release_device(self._h), "sa_batch_release_device")

    def prepare(self):
        """sa_batch_prepare: a deferred batch's plan and launch lists now (while the batch before it runs)."""
        _chk(lib().sa_batch_prepare(self._h), "sa_batch_prepare")

    def start(self):
        """sa_batch_start: run on a library thread; wait() joins it."""
        _chk(lib().sa_batch_start(self._h), "sa_batch_start")

    def wait(self):
        _chk(lib().sa_batch_wait(self._h), "sa_batch_wait")

    def pairs(self, job):
        n = C.c_int64()
        _chk(lib().sa_batch_n_pairs(self._h, job, C.byref(n)), "sa_batch_n_pairs")
        out = np.zeros(n.value, dtype=PAIR_DTYPE)
        if n.value:
            _chk(lib().sa_batch_pairs(self._h, job, out.ctypes.data, n.value), "sa_batch_pairs")
        return out

    def pairs16(self, job):
        """sa_batch_pairs16: the job's packed 16-byte records in place (a view into the batch's pinned result block, valid until
        the batch runs again or is closed): uint64 array [n, 2] of (a, b), see sa_pair16_t."""
        n, ptr = C.c_int64(), C.c_void_p()
        _chk(lib().sa_batch_pairs16(self._h, job, C.byref(ptr), C.byref(n)), "sa_batch_pairs16")
        if n.value == 0:
            return np.zeros((0, 2), dtype=np.uint64)
        buf = (C.c_uint64 * (2 * n.value)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 2)

    def pairs8(self, job):
        """sa_batch_pairs8 (a FLAG_PAIRS8 batch): the job's records decoded into a structured array (x, y, prob_e7)"""
        ptr, n = C.c_void_p(), C.c_int64()
        _chk(lib().sa_batch_pairs8(self._h, job, C.byref(ptr), C.byref(n)), "sa_batch_pairs8")
        out = np.zeros(n.value, dtype=[("x", np.int32), ("y", np.int32), ("prob_e7", np.int64)])
        if n.value:
            r = np.frombuffer((C.c_uint64 * n.value).from_address(ptr.value), dtype=np.uint64)
            out["x"] = (r & np.uint64(0xfffff)).astype(np.int32)
            out["y"] = ((r >> np.uint64(20)) & np.uint64(0xfffff)).astype(np.int32)
            out["prob_e7"] = (r >> np.uint64(40)).astype(np.int64)
        return out

    def results_view(self, first=None):
        """sa_batch_pairs16_all: what a finished batch holds, without copying anything -- (uint64 view [total, 2] of every job's
        packed records, first) with job j's records at view[first[j]:first[j + 1]].  `first`: an int64 array of n_jobs + 1
        entries to reuse.  (A FLAG_PAIRS8 batch: sa_batch_pairs8_all, a view [total, 1].)"""
        if first is None:
            first = np.zeros(self.n_jobs + 1, dtype=np.int64)
        ptr = C.c_void_p()
        if self._flags & FLAG_PAIRS8:
            _chk(lib().sa_batch_pairs8_all(self._h, C.byref(ptr), _ip(first)), "sa_batch_pairs8_all")
            total = int(first[self.n_jobs])
            if total == 0:
                return np.zeros((0, 1), dtype=np.uint64), first
            return np.frombuffer((C.c_uint64 * total).from_address(ptr.value), dtype=np.uint64).reshape(-1, 1), first
        _chk(lib().sa_batch_pairs16_all(self._h, C.byref(ptr), _ip(first)), "sa_batch_pairs16_all")
        total = int(first[self.n_jobs])
        if total == 0:
            return np.zeros((0, 2), dtype=np.uint64), first
        buf = (C.c_uint64 * (2 * total)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 2), first

    def pairs_all(self, out=None):
        """sa_batch_pairs_all: every job's pairs as sa_pair_t, expanded on the library's host threads.  Returns (rows, first)
        with rows[first[j]:first[j + 1]] job j's.  `out`: a PAIR_DTYPE array to reuse (large enough)."""
        first = np.zeros(self.n_jobs + 1, dtype=np.int64)
        rc = lib().sa_batch_pairs_all(self._h, None, 0, _ip(first))
        if rc not in (0, -1):
            _chk(rc, "sa_batch_pairs_all")
        total = int(first[-1])
        if out is None or len(out) < total:
            out = np.empty(total, dtype=PAIR_DTYPE)
        _chk(lib().sa_batch_pairs_all(self._h, out.ctypes.data, len(out), _ip(first)), "sa_batch_pairs_all")
        return out[:total], first

    def all_pairs_summary(self, job):
        """(number of pairs, sum of their prob_e7) over every pair above the threshold -- with FLAG_VC_ROWS including the rows the
        device dropped"""
        n, s = C.c_int64(0), C.c_int64(0)
        _chk(lib().sa_batch_all_pairs_summary(self._h, job, C.byref(n), C.byref(s)), "sa_batch_all_pairs_summary")
        return int(n.value), int(s.value)

    def n_pairs(self, job):
        n = C.c_int64()
        _chk(lib().sa_batch_n_pairs(self._h, job, C.byref(n)), "sa_batch_n_pairs")
        return n.value

    def mea(self, stats=None):
        """sa_batch_mea: the maximum-expected-accuracy path of every read of this (finished) batch, built from the pairs
        that are still on the device.  Returns per job (path [n, 2] of (x, y), best sum, status)."""
        n = self.n_jobs
        ptrs = (C.c_void_p * max(n, 1))()
        cnt = np.zeros(max(n, 1), dtype=np.int64)
        sums = np.zeros(max(n, 1), dtype=np.float64)
        st = np.zeros(max(n, 1), dtype=np.int32)
        kms = C.c_double()
        t0 = time.perf_counter()
        _chk(lib().sa_batch_mea(self._h, 0, ptrs, _ip(cnt), _dp(sums), st.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(kms)),
             "sa_batch_mea")
        if stats is not None:
            stats["kernel_ms"] = kms.value
            stats["call_ms"] = (time.perf_counter() - t0) * 1e3
        out = []
        for i in range(n):
            a = np.zeros((int(cnt[i]), 2), dtype=np.int32)
            if cnt[i]:
                C.memmove(a.ctypes.data, ptrs[i], 8 * int(cnt[i]))
            lib().sa_free(ptrs[i])
            out.append((a, float(sums[i]), int(st[i])))
        return out

    def stats(self):
        s = BatchStats()
        _chk(lib().sa_batch_stats(self._h, C.byref(s)), "sa_batch_stats")
        return s

    def close(self):
        if self._h:
            lib().sa_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def expect_batch(model, params, jobs, ambig=None, device=0, flags=0, pseudocount=0.0):
    """sa_expect_batch: per job the 3x3 transition expectations (from*3+to), the summed log-likelihood and the HDP
    assignments as (reference position, event index) arrays."""
    if isinstance(jobs, JobArray):
        n, arr, keep = jobs.n, jobs.arr, jobs
    else:
        n = len(jobs)
        arr, keep = _make_jobs(jobs)
    amb = ambig if ambig is not None else default_ambig()
    trans = np.full((max(n, 1), 9), pseudocount, dtype=np.float64)
    lik = np.zeros(max(n, 1), dtype=np.float64)
    ptrs = (C.c_void_p * max(n, 1))()
    cnt = np.zeros(max(n, 1), dtype=np.int64)
    _chk(lib().sa_expect_batch(model._h, C.byref(params), arr, n, amb, device, flags, _dp(trans), _dp(lik), ptrs,
                               _ip(cnt)), "sa_expect_batch")
    assigns = []
    for j in range(n):
        a = np.zeros((int(cnt[j]), 2), dtype=np.int64)
        if cnt[j]:
            C.memmove(a.ctypes.data, ptrs[j], 16 * int(cnt[j]))
        lib().sa_free(ptrs[j])
        assigns.append(a)
    del keep
    return trans[:n], lik[:n], assigns


def expect_last_stats():
    """sa_expect_last_stats: BatchStats of this thread's last expect_batch call."""
    s = BatchStats()
    _chk(lib().sa_expect_last_stats(C.byref(s)), "sa_expect_last_stats")
    return s


def plan_describe(model, params, job, ambig=None, flags=0):
    """Host-only view of the plan of ONE job: regions, band rows, traceback segments (no GPU needed)."""
    arr, keep = _make_jobs([job])
    amb = ambig if ambig is not None else default_ambig()
    info = PlanInfo()
    nev = arr[0].n_events
    lx = max(arr[0].ref_len, 0)
    cap_rows = int(lx + nev + 8 * (arr[0].n_anchors + 4))
    regions = np.zeros(4 * (arr[0].n_anchors + 2), dtype=np.int64)
    rows = np.zeros(3 * cap_rows, dtype=np.int64)
    segs = np.zeros(4 * (cap_rows // 100 + 16), dtype=np.int64)
    _chk(lib().sa_plan_describe(model._h, C.byref(params), arr, amb, flags, C.byref(info), _ip(regions),
                                len(regions) // 4, _ip(rows), cap_rows, _ip(segs), len(segs) // 4), "sa_plan_describe")
    nrow = 0
    reg = regions[:4 * info.n_regions].reshape(-1, 4)
    for r in reg:
        nrow += (r[2] - r[0]) + (r[3] - r[1]) + 1
    return info, reg, rows[:3 * nrow].reshape(-1, 3), segs[:4 * info.n_segments].reshape(-1, 4)


FLAG_RNA = 4


def scalings_mom(model, sequence, event_means, flags=0):
    """sa_scalings_mom: (shift, scale) by the method of moments (impl/eventAligner.c:784-843)."""
    ev = np.ascontiguousarray(event_means, dtype=np.float64)
    sb = sequence.encode()
    sh, sc = C.c_double(), C.c_double()
    _chk(lib().sa_scalings_mom(model._h, sb, len(sb), _dp(ev), len(ev), flags, C.byref(sh), C.byref(sc)), "sa_scalings_mom")
    return sh.value, sc.value


def event_align_batch(model, jobs, device=0, flags=0, stats=None):
    """sa_event_align_batch.  jobs: dicts(sequence, event_mean, scale, shift, var=1).  Returns per job
    (kmer_idx array, event_idx array, status); stats (a dict, optional) receives cells per job and the kernel time."""
    n = len(jobs)
    arr = (EaJob * max(n, 1))()
    keep = []
    for i, j in enumerate(jobs):
        ev = np.ascontiguousarray(j["event_mean"], dtype=np.float64)
        sb = j["sequence"].encode()
        keep.append((ev, sb))
        arr[i] = EaJob(sb, len(sb), _dp(ev), len(ev), j["scale"], j["shift"], j.get("var", 1.0))
    ptrs = (C.c_void_p * max(n, 1))()
    cnt = np.zeros(max(n, 1), dtype=np.int64)
    st = np.zeros(max(n, 1), dtype=np.int32)
    cells = np.zeros(max(n, 1), dtype=np.float64)
    kms = C.c_double()
    t0 = time.perf_counter()
    _chk(lib().sa_event_align_batch(model._h, arr, n, device, flags, ptrs, _ip(cnt), st.ctypes.data_as(C.POINTER(C.c_int32)),
                                    _dp(cells), C.byref(kms)), "sa_event_align_batch")
    if stats is not None:
        stats["cells"] = cells[:n].copy()
        stats["kernel_ms"] = kms.value
        stats["call_ms"] = (time.perf_counter() - t0) * 1e3  # the C call alone, without this wrapper's marshalling
    out = []
    for i in range(n):
        a = np.zeros((int(cnt[i]), 2), dtype=np.int32)
        if cnt[i]:
            C.memmove(a.ctypes.data, ptrs[i], 8 * int(cnt[i]))
        lib().sa_free(ptrs[i])
        out.append((a[:, 0].copy(), a[:, 1].copy(), int(st[i])))
    del keep
    return out


def _i32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def mea_batch(jobs, device=0, flags=0, stats=None):
    """sa_mea_batch.  jobs: dicts(event_idx, ref_idx, posterior, shortest) -- the COO posterior matrix and
    shortest_ref_per_event (np.inf or MEA_INF for events without rows).  Returns per job (path [n, 2] of (ref, event),
    best sum, status, number of final forward edges)."""
    n = len(jobs)
    arr = (MeaJob * max(n, 1))()
    keep = []
    for i, j in enumerate(jobs):
        ev = np.ascontiguousarray(j["event_idx"], dtype=np.int32)
        rf = np.ascontiguousarray(j["ref_idx"], dtype=np.int32)
        po = np.ascontiguousarray(j["posterior"], dtype=np.float64)
        sh = np.asarray(j["shortest"], dtype=np.float64)
        sh = np.ascontiguousarray(np.where(np.isfinite(sh), sh, MEA_INF).astype(np.int32))
        keep.append((ev, rf, po, sh))
        arr[i] = MeaJob(_i32p(ev), _i32p(rf), _dp(po), len(ev), _i32p(sh), len(sh))
    ptrs = (C.c_void_p * max(n, 1))()
    cnt = np.zeros(max(n, 1), dtype=np.int64)
    sums = np.zeros(max(n, 1), dtype=np.float64)
    st = np.zeros(max(n, 1), dtype=np.int32)
    ne = np.zeros(max(n, 1), dtype=np.int32)
    kms = C.c_double()
    t0 = time.perf_counter()
    _chk(lib().sa_mea_batch(arr, n, device, flags, ptrs, _ip(cnt), _dp(sums), _i32p(st), _i32p(ne), C.byref(kms)), "sa_mea_batch")
    if stats is not None:
        stats["kernel_ms"] = kms.value
        stats["call_ms"] = (time.perf_counter() - t0) * 1e3
    out = []
    for i in range(n):
        a = np.zeros((int(cnt[i]), 2), dtype=np.int32)
        if cnt[i]:
            C.memmove(a.ctypes.data, ptrs[i], 8 * int(cnt[i]))
        lib().sa_free(ptrs[i])
        out.append((a, float(sums[i]), int(st[i]), int(ne[i])))
    del keep
    return out


def mea_params(reference_index, event_index, posterior):
    """sa_mea_params: event-table columns -> (event_idx, ref_idx, posterior, shortest_ref_per_event)."""
    ri = np.ascontiguousarray(reference_index, dtype=np.int64)
    ei = np.ascontiguousarray(event_index, dtype=np.int64)
    po = np.ascontiguousarray(posterior, dtype=np.float64)
    n = len(ri)
    rows = np.zeros(max(n, 1), dtype=np.int32)
    cols = np.zeros(max(n, 1), dtype=np.int32)
    data = np.zeros(max(n, 1), dtype=np.float64)
    sh = np.zeros(int(ei.max() - ei.min() + 1) if n else 1, dtype=np.int32)
    ne = C.c_int64()
    m = lib().sa_mea_params(_ip(ri), _ip(ei), _dp(po), n, _i32p(rows), _i32p(cols), _dp(data), _i32p(sh), C.byref(ne))
    if m < 0:
        _chk(int(m), "sa_mea_params")
    return rows[:m].copy(), cols[:m].copy(), data[:m].copy(), sh


def plan_digest(model, params, jobs, ambig=None, flags=0, threads=0):
    """Host-only: plan a batch with `threads` planner threads; returns (PlanInfo, digest of everything uploaded)."""
    arr, keep = _make_jobs(jobs)
    amb = ambig if ambig is not None else default_ambig()
    info, dig = PlanInfo(), C.c_uint64()
    _chk(lib().sa_plan_digest(model._h, C.byref(params), arr, len(jobs), amb, flags, threads, C.byref(info), C.byref(dig)),
         "sa_plan_digest")
    del keep
    return info, dig.value


def guide_to_anchors(start1, end1, strand1, start2, ops, trim):
    t = np.array([o[0] for o in ops], dtype=np.int32)
    ln = np.array([o[1] for o in ops], dtype=np.int64)
    cap = int(ln.sum()) + 1
    ax, ay = np.zeros(cap, dtype=np.int64), np.zeros(cap, dtype=np.int64)
    n = lib().sa_guide_to_anchors(start1, end1, int(strand1), start2, t.ctypes.data_as(C.POINTER(C.c_int32)), _ip(ln),
                                  len(t), trim, _ip(ax), _ip(ay), cap)
    if n < 0:
        raise SaError(int(n), "sa_guide_to_anchors")
    return ax[:n].copy(), ay[:n].copy()


def remap_anchors(ax, ay, event_map, map_offset):
    axa = np.ascontiguousarray(ax, dtype=np.int64)
    aya = np.ascontiguousarray(ay, dtype=np.int64)
    em = np.ascontiguousarray(event_map, dtype=np.int64)
    ox, oy = np.zeros(len(axa) + 1, dtype=np.int64), np.zeros(len(axa) + 1, dtype=np.int64)
    n = lib().sa_remap_anchors(_ip(axa), _ip(aya), len(axa), _ip(em), map_offset, _ip(ox), _ip(oy))
    if n < 0:
        raise SaError(int(n), "sa_remap_anchors")
    return ox[:n].copy(), oy[:n].copy()


def estimate_params(model, table5, strand_event_map, events4, strand_read):
    em = np.ascontiguousarray(strand_event_map, dtype=np.int64)
    out = np.zeros(7, dtype=np.float64)
    _chk(lib().sa_estimate_params(model._h, _dp(table5), _ip(em), _dp(events4), events4.shape[0],
                                  strand_read.encode(), len(strand_read), _dp(out)), "sa_estimate_params")
    return dict(zip(["scale", "shift", "var", "drift", "scale_sd", "var_sd", "shift_sd"], out.tolist()))


def plan_check_path_records(model, params, job, ambig=None):
    """Host-only test hook: (wrong entries, entries checked) of the per-path neighbour records of `job`'s plan."""
    arr, keep = _make_jobs([job])
    amb = ambig if ambig is not None else default_ambig()
    n = C.c_int64(0)
    bad = lib().sa_plan_check_path_records(model._h, C.byref(params), arr, amb, C.byref(n))
    if bad < 0:
        _chk(int(bad), "sa_plan_check_path_records")
    return int(bad), int(n.value)


def dplan_compare(model, params, jobs, ambig=None, device=0, flags=0):
    """Test hook (GPU): 0 when the device planner and the host planner produce identical arrays for `jobs`."""
    if isinstance(jobs, JobArray):
        arr, keep, n = jobs.arr, jobs, jobs.n
    else:
        arr, keep = _make_jobs(jobs)
        n = len(jobs)
    amb = ambig if ambig is not None else default_ambig()
    rc = lib().sa_dplan_compare(model._h, C.byref(params), arr, n, amb, device, flags)
    del keep
    if rc < 0:
        _chk(rc, "sa_dplan_compare")
    return rc


def pool_configure(device_limit_bytes=-1, pinned_limit_bytes=-1):
    """Bounds of what the caching allocators may keep parked between batches (sa_pool_configure); -1 leaves a bound as it is."""
    _chk(lib().sa_pool_configure(int(device_limit_bytes), int(pinned_limit_bytes)), "sa_pool_configure")


def device_memory(device=0):
    """(free bytes, total bytes) of the GPU's HBM; what the library's caching allocator holds counts as free."""
    f, t = C.c_int64(), C.c_int64()
    _chk(lib().sa_device_memory(device, C.byref(f), C.byref(t)), "sa_device_memory")
    return f.value, t.value


HMM_GAUSSIAN, HMM_HDP = 0, 1


class Hmm:
    """sa_hmm_t: the expectations object of the EM loop (Hmm / ContinuousPairHmm / HdpHmm, impl/continuousHmm.c).  The numpy arrays
    `transitions`, `event_model`, `event_expectations`, `posteriors`, `observed` are VIEWS of the object's own storage."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def create(cls, model, kind=HMM_GAUSSIAN, threshold=0.0, transitions_pseudocount=0.0, emissions_pseudocount=0.0):
        h = C.c_void_p()
        _chk(lib().sa_hmm_create(C.byref(h), model._h, kind, threshold, transitions_pseudocount, emissions_pseudocount),
             "sa_hmm_create")
        return cls(h)

    @classmethod
    def load(cls, path, kind=HMM_GAUSSIAN, transitions_pseudocount=0.0, emissions_pseudocount=0.0):
        h = C.c_void_p()
        _chk(lib().sa_hmm_load(C.byref(h), os.fsencode(path), kind, transitions_pseudocount, emissions_pseudocount), "sa_hmm_load")
        return cls(h)

    def view(self):
        v = HmmView()
        _chk(lib().sa_hmm_view(self._h, C.byref(v)), "sa_hmm_view")
        return v

    def _arr(self, ptr, n, dtype=np.float64):
        if not ptr or n == 0:
            return np.zeros(0, dtype=dtype)
        return np.ctypeslib.as_array(ptr, shape=(int(n),))

    @property
    def transitions(self):
        return self._arr(self.view().transitions, 9).reshape(3, 3)

    @property
    def likelihood(self):
        return float(self.view().likelihood[0])

    @property
    def event_model(self):
        v = self.view()
        return self._arr(v.event_model, 5 * v.n_kmers).reshape(-1, 5)

    @property
    def event_expectations(self):
        v = self.view()
        return self._arr(v.event_expectations, 2 * v.n_kmers).reshape(-1, 2)

    @property
    def posteriors(self):
        v = self.view()
        return self._arr(v.posteriors, v.n_kmers)

    @property
    def observed(self):
        v = self.view()
        return self._arr(v.observed, v.n_kmers, np.uint8)

    def assignments(self):
        """(k-mers as a list of str, event means as an array) of an HdpHmm"""
        v = self.view()
        n, k = int(v.n_assignments), int(v.k)
        if n == 0:
            return [], np.zeros(0)
        raw = C.string_at(v.assignment_kmers, n * k).decode()
        return [raw[i * k:(i + 1) * k] for i in range(n)], np.ctypeslib.as_array(v.assignment_events, shape=(n,)).copy()

    def set_event_model(self, table5):
        t = np.ascontiguousarray(table5, dtype=np.float64)
        _chk(lib().sa_hmm_set_event_model(self._h, _dp(t)), "sa_hmm_set_event_model")

    def add_expectations(self, trans9, likelihood):
        t = np.ascontiguousarray(trans9, dtype=np.float64).reshape(-1)
        _chk(lib().sa_hmm_add_expectations(self._h, _dp(t), float(likelihood)), "sa_hmm_add_expectations")

    def add_emission_expectation(self, kmer_index, mean, p):
        _chk(lib().sa_hmm_add_emission_expectation(self._h, int(kmer_index), float(mean), float(p)), "sa_hmm_add_emission_expectation")

    def add_assignment(self, kmer, event_mean):
        _chk(lib().sa_hmm_add_assignment(self._h, kmer.encode(), float(event_mean)), "sa_hmm_add_assignment")

    def write(self, path):
        _chk(lib().sa_hmm_write(self._h, os.fsencode(path)), "sa_hmm_write")

    def add_expectations_file(self, path):
        """HMM.add_expectations_file: a read's .expectations file added to this object's accumulators"""
        _chk(lib().sa_hmm_add_expectations_file(self._h, os.fsencode(path)), "sa_hmm_add_expectations_file")

    def normalize(self):
        _chk(lib().sa_hmm_normalize(self._h), "sa_hmm_normalize")

    def load_into_model(self, model):
        _chk(lib().sa_hmm_load_into_model(model._h, self._h), "sa_hmm_load_into_model")

    def close(self):
        if self._h:
            lib().sa_hmm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HdpState:
    """The whole state of a serialised NanoporeHDP (sa_hdp_state_*: the deterministic pieces of the HDP rebuild).  Arrays are
    copies (numpy) of the views sa_hdp_state_info hands out."""

    def __init__(self, path=None, _handle=None):
        self._h = _handle if _handle is not None else C.c_void_p()
        if _handle is None:
            _chk(lib().sa_hdp_state_load(C.byref(self._h), os.fsencode(path)), "sa_hdp_state_load")
        self.info = HdpStateInfo()
        self.refresh()

    def refresh(self):
        """the info views again (after a call that changed the state: data passed, a sampling run, finalisation)"""
        _chk(lib().sa_hdp_state_info(self._h, C.byref(self.info)), "sa_hdp_state_info")

    @classmethod
    def new(cls, layout, alphabet, kmer_length, grid, nig, gamma=None, gamma_alpha=None, gamma_beta=None, groups=None):
        """sa_hdp_state_new: a NanoporeHDP without data.  layout: HDP_LAYOUT_*; grid = (start, stop, length); nig = (mu, nu, alpha,
        beta); gamma: fixed concentration parameters by depth, or gamma_alpha / gamma_beta: a Gamma prior on them."""
        h = C.c_void_p()
        g = None if gamma is None else np.ascontiguousarray(gamma, dtype=np.float64)
        ga = None if gamma_alpha is None else np.ascontiguousarray(gamma_alpha, dtype=np.float64)
        gb = None if gamma_beta is None else np.ascontiguousarray(gamma_beta, dtype=np.float64)
        gr = None if groups is None else np.ascontiguousarray(groups, dtype=np.int64)
        depth = 2 if int(layout) == HDP_LAYOUT_FLAT else 3   # (the C entry point reads `depth` values of each vector it is given)
        for v in (g, ga, gb):
            if v is not None and len(v) != depth:
                raise SaError(-1, "sa_hdp_state_new: %d concentration parameters for a layout of depth %d" % (len(v), depth))
        if gr is not None and len(gr) != len(alphabet):
            raise SaError(-1, "sa_hdp_state_new: one group per letter of the alphabet")
        _chk(lib().sa_hdp_state_new(C.byref(h), int(layout), alphabet.encode(), int(kmer_length), None if gr is None else _ip(gr),
                                    None if g is None else _dp(g), None if ga is None else _dp(ga), None if gb is None else _dp(gb),
                                    float(grid[0]), float(grid[1]), int(grid[2]), float(nig[0]), float(nig[1]), float(nig[2]), float(nig[3])),
             "sa_hdp_state_new")
        return cls(_handle=h)

    @classmethod
    def new_tree(cls, parents, depth, grid, nig, gamma=None, gamma_alpha=None, gamma_beta=None):
        """sa_hdp_state_new_tree: a plain HierarchicalDirichletProcess over the tree `parents` (-1 for the base DP)"""
        h = C.c_void_p()
        pa = np.ascontiguousarray(parents, dtype=np.int64)
        g = None if gamma is None else np.ascontiguousarray(gamma, dtype=np.float64)
        ga = None if gamma_alpha is None else np.ascontiguousarray(gamma_alpha, dtype=np.float64)
        gb = None if gamma_beta is None else np.ascontiguousarray(gamma_beta, dtype=np.float64)
        for v in (g, ga, gb):
            if v is not None and len(v) != int(depth):
                raise SaError(-1, "sa_hdp_state_new_tree: %d concentration parameters for depth %d" % (len(v), int(depth)))
        _chk(lib().sa_hdp_state_new_tree(C.byref(h), len(pa), int(depth), _ip(pa), None if g is None else _dp(g),
                                         None if ga is None else _dp(ga), None if gb is None else _dp(gb), float(grid[0]), float(grid[1]),
                                         int(grid[2]), float(nig[0]), float(nig[1]), float(nig[2]), float(nig[3])), "sa_hdp_state_new_tree")
        return cls(_handle=h)

    def pass_data(self, data, dp_ids):
        d = np.ascontiguousarray(data, dtype=np.float64)
        i = np.ascontiguousarray(dp_ids, dtype=np.int64)
        _chk(lib().sa_hdp_state_pass_data(self._h, _dp(d), _ip(i), len(d)), "sa_hdp_state_pass_data")
        self.refresh()

    def pass_assignments(self, kmers, events):
        """(k-mers as a list of str or one concatenated str, event means): hdpHmm_loadFromFile's hand-over"""
        km = kmers if isinstance(kmers, str) else "".join(kmers)
        e = np.ascontiguousarray(events, dtype=np.float64)
        _chk(lib().sa_hdp_state_pass_assignments(self._h, km.encode(), _dp(e), len(e)), "sa_hdp_state_pass_assignments")
        self.refresh()

    def pass_assignment_file(self, path, strand=None):
        n = C.c_int64()
        _chk(lib().sa_hdp_state_pass_assignment_file(self._h, os.fsencode(path), None if strand is None else strand.encode(), C.byref(n)),
             "sa_hdp_state_pass_assignment_file")
        self.refresh()
        return n.value

    def kmer_dp(self, kmer):
        return lib().sa_hdp_state_kmer_dp(self._h, kmer.encode())

    def gibbs(self, num_samples, burn_in, thinning, seed=1, device=0, verbose=False):
        _chk(lib().sa_hdp_state_gibbs(self._h, int(num_samples), int(burn_in), int(thinning), int(seed), device, 1 if verbose else 0),
             "sa_hdp_state_gibbs")
        self.refresh()

    def finalize(self, device=0):
        _chk(lib().sa_hdp_state_finalize(self._h, device), "sa_hdp_state_finalize")
        self.refresh()

    def samples_taken(self):
        return int(lib().sa_hdp_state_samples_taken(self._h))

    def close(self):
        if self._h:
            lib().sa_hdp_state_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def write(self, path):
        _chk(lib().sa_hdp_state_write(self._h, os.fsencode(path)), "sa_hdp_state_write")

    def array(self, name):
        """a copy of one of the info views, shaped: per data point, per DP, per factor (f_params: n_factors x 5), post / slope:
        n_observed x grid_length"""
        i = self.info
        n = {"data": i.n_data, "data_dp": i.n_data, "gamma": i.depth, "grid": i.grid_length, "dp_parent": i.num_dps,
             "dp_num_factor_children": i.num_dps, "dp_depth": i.num_dps, "observed": i.num_dps, "row_of_dp": i.num_dps,
             "post": i.n_observed * i.grid_length, "slope": i.n_observed * i.grid_length, "f_type": i.n_factors,
             "f_parent": i.n_factors, "f_ref": i.n_factors, "f_params": 5 * i.n_factors, "f_n_children": i.n_factors}[name]
        p = getattr(i, name)
        a = np.ctypeslib.as_array(p, shape=(int(n),)).copy() if n > 0 else np.zeros(0)
        if name in ("post", "slope"):
            a = a.reshape(int(i.n_observed), int(i.grid_length))
        if name == "f_params":
            a = a.reshape(int(i.n_factors), 5)
        return a

    def sample_weights(self):
        """sa_hdp_state_sample_weights (host code): (row_start, col, w) of one sample's weights, CSR over the observed DPs' rows"""
        rs, col, w, nnz = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.POINTER(C.c_double)(), C.c_int64()
        _chk(lib().sa_hdp_state_sample_weights(self._h, C.byref(rs), C.byref(col), C.byref(w), C.byref(nnz)),
             "sa_hdp_state_sample_weights")
        n = int(nnz.value)
        out = (np.ctypeslib.as_array(rs, shape=(int(self.info.n_observed) + 1,)).copy(),
               np.ctypeslib.as_array(col, shape=(max(n, 1),))[:n].copy(), np.ctypeslib.as_array(w, shape=(max(n, 1),))[:n].copy())
        for q in (rs, col, w):
            lib().sa_free(C.cast(q, C.c_void_p))
        return out

    def distr_sample(self, device=0):
        """sa_hdp_state_distr_sample: what one sample of this state adds to every observed DP's collector (GPU)"""
        out = np.zeros((int(self.info.n_observed), int(self.info.grid_length)), dtype=np.float64)
        _chk(lib().sa_hdp_state_distr_sample(self._h, device, _dp(out)), "sa_hdp_state_distr_sample")
        return out


HDP_LAYOUT_FLAT, HDP_LAYOUT_MULTISET, HDP_LAYOUT_MIDDLE_NTS, HDP_LAYOUT_COMPOSITION, HDP_LAYOUT_GROUP_MULTISET = 0, 1, 2, 3, 4


def hdp_nig_params_from_table(table5):
    """sa_hdp_nig_params_from_table: (mu, nu, alpha, beta) by maximum likelihood from a lookup table (5 doubles per k-mer)"""
    t = np.ascontiguousarray(table5, dtype=np.float64).reshape(-1)
    out = [C.c_double() for _ in range(4)]
    _chk(lib().sa_hdp_nig_params_from_table(_dp(t), len(t) // 5, *[C.byref(o) for o in out]), "sa_hdp_nig_params_from_table")
    return tuple(o.value for o in out)


def hdp_finalize_distributions(grid, collectors, samples, device=0):
    """sa_hdp_finalize_distributions: (collectors / samples, spline slopes), both n_rows x grid_length (GPU)"""
    grid = np.ascontiguousarray(grid, dtype=np.float64)
    s = np.ascontiguousarray(collectors, dtype=np.float64).reshape(-1, len(grid))
    y, k = np.zeros_like(s), np.zeros_like(s)
    _chk(lib().sa_hdp_finalize_distributions(_dp(grid), len(grid), _dp(s), s.shape[0], int(samples), device, _dp(y), _dp(k)),
         "sa_hdp_finalize_distributions")
    return y, k
