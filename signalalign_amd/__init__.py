"""signalalign_amd -- MI355X-native banded pair-HMM path of signalAlign.

The product is the C-ABI shared library `lib/libsignalalign_hip.so` (see include/signalalign_hip.h) and
the `bin/signalMachine` drop-in CLI; this package is a thin ctypes mirror of that ABI for tests, the
benchmark and Python callers.  There is no CPU fallback: without the built library, or without a GPU
for the compute calls, everything here raises.
"""
from ._capi import (Batch, JobArray, Model, Params, SaError, build, default_ambig, default_params, device_count, device_memory, pool_configure, lib,
                    library_path, plan_describe, plan_digest, plan_check_path_records, dplan_compare, expect_batch, expect_last_stats, scalings_mom, event_align_batch, mea_batch, mea_params, MEA_INF, guide_to_anchors, remap_anchors, estimate_params, PAIR_DTYPE,
                    FLAG_EXACT, FLAG_FORCE_GENERIC, FLAG_RNA, FLAG_DEVICE_TO_ITSELF, FLAG_VC_ROWS, FLAG_PAIRS8,
                    FLAG_INPUTS_IN_HOST_BLOCK, HostBlock, HdpState, hdp_finalize_distributions, Hmm, HMM_GAUSSIAN, HMM_HDP, hdp_nig_params_from_table,
                    HDP_LAYOUT_FLAT, HDP_LAYOUT_MULTISET, HDP_LAYOUT_MIDDLE_NTS, HDP_LAYOUT_COMPOSITION, HDP_LAYOUT_GROUP_MULTISET)

__all__ = ["Batch", "JobArray", "Model", "Params", "SaError", "build", "default_ambig", "default_params", "device_count", "device_memory", "pool_configure", "lib",
           "library_path", "plan_describe", "plan_digest", "plan_check_path_records", "dplan_compare", "expect_batch", "expect_last_stats", "scalings_mom", "event_align_batch", "mea_batch", "mea_params", "MEA_INF", "guide_to_anchors", "remap_anchors", "estimate_params", "PAIR_DTYPE",
           "FLAG_EXACT", "FLAG_FORCE_GENERIC", "FLAG_RNA", "FLAG_DEVICE_TO_ITSELF", "FLAG_VC_ROWS", "FLAG_PAIRS8",
           "FLAG_INPUTS_IN_HOST_BLOCK", "HostBlock", "HdpState", "hdp_finalize_distributions", "Hmm", "HMM_GAUSSIAN", "HMM_HDP", "hdp_nig_params_from_table",
           "HDP_LAYOUT_FLAT", "HDP_LAYOUT_MULTISET", "HDP_LAYOUT_MIDDLE_NTS", "HDP_LAYOUT_COMPOSITION", "HDP_LAYOUT_GROUP_MULTISET"]
