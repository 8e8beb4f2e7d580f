"""ctypes binding of libbgseq.so (include/bgseq.h).

The library is the product: HIP kernels for gfx950 behind a C ABI.  There is no CPU fallback — if
the shared library is missing, or no HIP device is usable, calls raise.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BG_LIBBGSEQ: load another build of the library (timing experiments, tools/variant_times.sh) without touching the real one
LIB_PATH = os.environ.get("BG_LIBBGSEQ") or os.path.join(_HERE, "libbgseq.so")
_lib = None

BG_OK = 0
ERRORS = {-1: "BG_EINVAL", -2: "BG_ENODEV", -3: "BG_ENOMEM", -4: "BG_EBADBASE", -5: "BG_EIO", -6: "BG_EINTERNAL"}


class BgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERRORS.get(code, code), msg))
        self.code = code


class Timing(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_double), ("launches", C.c_uint64), ("bytes", C.c_uint64)]


class CorrectParams(C.Structure):
    _fields_ = [("max_corrections", C.c_uint32), ("min_good_run", C.c_uint32), ("trim_after_portion", C.c_float)]


class CorrectStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("corrected_reads", "corrected_bases", "failed", "modified",
                                          "modified_bases", "truncated", "dropped_bases")]


class BuildStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_reads", "n_bases", "n_kmers", "n_distinct_kmers", "n_kept_kmers",
                                          "n_corrected_reads", "n_corrected_bases", "n_records", "n_tied_records",
                                          "n_entries")] + [("max_len", C.c_uint32)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class ShardedStats(C.Structure):  # bg_sharded_stats
    _fields_ = [("build", BuildStats)] + [(n, C.c_uint64) for n in ("shard_entries", "first_entry", "slice_entries", "n_read_ids",
                                                                       "first_read_id", "slice_read_ids", "correct_probes",
                                                                       "tails_sent_in_full")] + [("correct_steps", C.c_uint32)]

    def as_dict(self):
        d = self.build.as_dict()
        d.update({n: int(getattr(self, n)) for n, _ in self._fields_[1:]})
        return d


class CommStats(C.Structure):  # bg_comm_stats
    _fields_ = [("collectives", C.c_uint64), ("bytes_through_backend", C.c_uint64), ("bytes_kept_local", C.c_uint64),
                ("seconds_in_calls", C.c_double), ("backend", C.c_int), ("rccl_version", C.c_int)]


class CreateParams(C.Structure):
    _fields_ = [("kmer_size", C.c_int), ("min_kmer_count", C.c_uint32), ("correct", CorrectParams)]


def _header_symbols():
    """Every function include/bgseq.h declares: the C ABI is written down once, in the header (the pybind11 module
    biograph_amd._capi is the SDK seam; this ctypes mirror serves the drivers and the tests and names no symbol list of
    its own)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "bgseq.h")
    with open(hdr) as f:
        return sorted(set(re.findall(r"\b(bg_[a-z0-9_]+)\s*\(", f.read())))


SYMBOLS = _header_symbols()

PART_KMER_HASH, PART_PREFIX, PART_KEY_BOUNDS, PART_ROW_DEST = 0, 1, 2, 3


def lib():
    """Loads libbgseq.so (building it is `python -m biograph_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("biograph_amd: %s is missing — build it with `python -m biograph_amd.build` "
                          "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    # When torch is part of the process (bench, sharded driver, tests) its bundled HIP runtime must be the one
    # that gets loaded: with libbgseq's /opt/rocm runtime in first, torch.cuda finds "no HIP GPUs".
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    P = C.POINTER
    L.bg_abi_version.restype = i32
    L.bg_last_error.restype = C.c_char_p
    L.bg_ctx_create.argtypes = [i32, vp, P(vp)]
    L.bg_ctx_destroy.argtypes = [vp]
    L.bg_ctx_destroy.restype = None
    L.bg_ctx_trim.argtypes = [vp]
    L.bg_ctx_timings.argtypes = [vp, P(Timing), i32]
    L.bg_ctx_set_profiling.argtypes = [vp, i32]
    L.bg_pack_reads.argtypes = [vp, vp, vp, u64, P(vp)]
    L.bg_pack_reads_host.argtypes = [vp, vp, vp, u64, P(vp)]
    for f in ("bg_reads_count", "bg_reads_total_bases", "bg_kmerset_size", "bg_kmerset_distinct",
              "bg_kmerset_instances", "bg_seqset_size"):
        getattr(L, f).restype = u64
        getattr(L, f).argtypes = [vp]
    L.bg_reads_max_len.restype = u32
    L.bg_reads_max_len.argtypes = [vp]
    L.bg_seqset_max_len.restype = u32
    L.bg_seqset_max_len.argtypes = [vp]
    L.bg_reads_download.argtypes = [vp, vp, vp, vp]
    for f in ("bg_reads_free", "bg_kmerset_free", "bg_seqset_free"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = None
    L.bg_kmer_count.argtypes = [vp, vp, i32, u32, P(vp)]
    L.bg_kmerset_download.argtypes = [vp, vp, vp, vp, vp]
    L.bg_correct_reads.argtypes = [vp, vp, vp, P(CorrectParams), P(vp), P(CorrectStats)]
    L.bg_seqset_build.argtypes = [vp, vp, i32, P(vp), P(BuildStats)]
    L.bg_create_params_default.argtypes = [P(CreateParams)]
    L.bg_create_params_default.restype = None
    L.bg_create.argtypes = [vp, vp, vp, u64, P(CreateParams), P(vp), P(BuildStats)]
    L.bg_seqset_part.argtypes = [vp, vp, C.c_char_p, vp, P(u64)]
    L.bg_seqset_arrays.argtypes = [vp, vp, vp, vp]
    L.bg_seqset_entries.argtypes = [vp, vp, u64, u64, vp, vp]
    L.bg_seqset_write.argtypes = [vp, vp, C.c_char_p, i32, P(C.c_char_p)]
    L.bg_seqset_open.argtypes = [vp, C.c_char_p, P(vp)]
    L.bg_seqset_from_arrays.argtypes = [vp, u64, vp, vp, P(vp), P(vp)]
    L.bg_push_front_batch.argtypes = [vp, vp, u64, vp, vp, vp, vp, vp, vp, vp]
    L.bg_push_front_drop_batch.argtypes = [vp, vp, u64, vp, vp, vp, vp, vp, vp, vp, vp]
    L.bg_pop_front_batch.argtypes = [vp, vp, u64, vp, vp, vp, vp, vp, vp]
    L.bg_find_batch.argtypes = [vp, vp, u64, vp, vp, vp, vp]
    L.bg_dev_alloc.argtypes = [vp, u64, P(vp)]
    L.bg_dev_free.argtypes = [vp, vp]
    L.bg_dev_upload.argtypes = [vp, vp, vp, u64]
    L.bg_dev_download.argtypes = [vp, vp, vp, u64]
    L.bg_trace_one_base_batch.argtypes = [vp, vp, u64] + [vp] * 15
    L.bg_branch_trace_batch.argtypes = [vp, vp, u64, vp, vp, vp, vp, vp, vp, u32, u32, vp, vp, vp, vp, vp, vp, vp]
    L.bg_ref_map_build.argtypes = [vp, vp, vp, u64, vp, vp, u32, vp]
    L.bg_ref_walk_entries.argtypes = [vp, vp, vp, u64, i32, vp]
    L.bg_tracer_extend_batch.argtypes = [vp, vp, vp, u64, vp, u32, vp, vp]
    L.bg_trace_options_default.argtypes = [vp]
    L.bg_trace_options_default.restype = None
    L.bg_push_trace.argtypes = [vp, vp, vp, vp, u64, vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, i32, i32, vp, P(vp)]
    L.bg_reference_build.argtypes = [C.c_char_p, C.c_char_p, u64]
    L.bg_bwt_build.argtypes = [vp, vp, u32, vp, vp, u64, C.c_char_p, vp]
    L.bg_tracer_open.argtypes = [vp, vp, vp, vp, vp, P(vp)]
    L.bg_tracer_clone.argtypes = [vp, vp, P(vp)]
    L.bg_ctx_create_own_stream.argtypes = [i32, P(vp)]
    L.bg_tracer_trace.argtypes = [vp, u64, vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, i32, i32, P(vp)]
    L.bg_tracer_trace_async.argtypes = [vp, u64, vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, i32, i32, P(vp)]
    L.bg_trace_result_wait.argtypes = [vp]
    L.bg_pop_trace.argtypes = [vp, u64, vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, i32, vp, u32, P(vp)]
    L.bg_trace_result_anchor_drops.argtypes = [vp]
    L.bg_trace_result_anchor_drops.restype = u64
    L.bg_push_front_chains.argtypes = [vp, vp, u32, vp, vp, vp]
    L.bg_tracer_close.argtypes = [vp]
    L.bg_tracer_close.restype = None
    L.bg_sha1_file.argtypes = [C.c_char_p, C.c_char_p]
    L.bg_trace_result_size.argtypes = [vp]
    L.bg_trace_result_size.restype = u64
    L.bg_trace_result_get.argtypes = [vp, u64, vp]
    L.bg_trace_result_stats.argtypes = [vp, vp]
    L.bg_trace_result_free.argtypes = [vp]
    L.bg_trace_result_free.restype = None
    L.bg_assemblies_process.argtypes = [vp, u32, u64, vp, vp, vp, C.c_int64, vp, u32, P(vp)]
    L.bg_pipeline_open.argtypes = [u64, vp, vp, vp, C.c_int64, vp, u32, P(vp)]
    L.bg_pipeline_feed.argtypes = [vp, vp, u32, C.c_int64, P(C.c_int)]
    L.bg_pipeline_finish.argtypes = [vp, P(vp)]
    L.bg_pipeline_take.argtypes = [vp, P(vp)]
    L.bg_pipeline_ready.argtypes = [vp]
    L.bg_pipeline_ready.restype = u64
    L.bg_pipeline_release_early.argtypes = [vp, C.c_int64]
    L.bg_pipeline_close.argtypes = [vp]
    L.bg_pipeline_close.restype = None
    L.bg_trace_result_new.argtypes = [P(vp)]
    L.bg_trace_result_add.argtypes = [vp, vp]
    L.bg_trace_result_half_aligned_count.argtypes = [vp]
    L.bg_trace_result_half_aligned_count.restype = u64
    L.bg_trace_result_half_aligned.argtypes = [vp, u64, vp, vp, vp, vp, vp]
    L.bg_trace_result_var_count.argtypes = [vp, u64]
    L.bg_trace_result_var_count.restype = u64
    L.bg_trace_result_var.argtypes = [vp, u64, u64, vp, vp, vp, vp]
    L.bg_coverage_options_default.argtypes = [vp]
    L.bg_coverage_options_default.restype = None
    L.bg_assemblies_coverage.argtypes = [vp, vp, vp, vp, u64, vp, vp, vp, C.c_int64, vp, i32, P(vp)]
    L.bg_coverage_open.argtypes = [vp, vp, vp, P(vp)]
    L.bg_coverage_run.argtypes = [vp, vp, u64, vp, vp, vp, C.c_int64, vp, i32, P(vp)]
    L.bg_coverage_close.argtypes = [vp]
    L.bg_coverage_close.restype = None
    L.bg_trace_result_coverage.argtypes = [vp, u64, vp]
    L.bg_trace_result_set_coverage.argtypes = [vp, u64, vp]
    L.bg_assemblies_rvg_exclude.argtypes = [vp, vp, P(vp)]
    L.bg_vcf_record.argtypes = [vp, u64, C.c_char_p, u64, vp, vp, vp, C.c_int64, vp, u64, vp, u64, P(u64)]
    L.bg_walk_chains.argtypes = [vp, vp, u32, vp, vp, u32, vp]
    L.bg_walk_chains_batch.argtypes = [vp, vp, u32, vp, vp, u64, vp, u64]
    L.bg_trace_result_ml.argtypes = [vp, u64, vp]
    L.bg_trace_result_set_ml.argtypes = [vp, u64, vp]
    L.bg_readmap_build.argtypes = [vp, vp, vp, i32, P(vp)]
    L.bg_readmap_build_marked.argtypes = [vp, vp, vp, i32, vp, P(vp)]
    L.bg_seqset_mark_inputs.argtypes = [vp, vp, vp, u64, u32, vp, P(u32)]
    L.bg_readmap_size.restype = u64
    L.bg_readmap_size.argtypes = [vp]
    L.bg_readmap_max_read_len.restype = u32
    L.bg_readmap_max_read_len.argtypes = [vp]
    L.bg_readmap_part.argtypes = [vp, vp, C.c_char_p, vp, P(u64)]
    L.bg_readmap_write.argtypes = [vp, vp, C.c_char_p, i32, P(C.c_char_p), C.c_char_p]
    L.bg_readmap_free.argtypes = [vp]
    L.bg_readmap_free.restype = None
    L.bg_readmap_open.argtypes = [vp, C.c_char_p, P(vp)]
    L.bg_readmap_arrays.argtypes = [vp, vp, vp, vp, vp, vp]
    L.bg_readmap_from_arrays.argtypes = [vp, u64, u32, C.c_char_p, u64, vp, vp, vp, vp, P(vp)]
    L.bg_cover_filter_build.argtypes = [vp, vp, i32, vp]
    L.bg_cover_query_scan.argtypes = [vp, vp, vp, i32, u32, u32, vp, vp, u64, vp, P(u64)]
    L.bg_cover_query_scan_answers.argtypes = [vp, vp, vp, vp, u64]
    L.bg_cover_query_records_split.argtypes = [vp, vp, i32, i32, u32, vp, vp, u64, vp, P(u64)]
    L.bg_cover_query_records_live.argtypes = [vp, vp, i32, P(u64)]
    L.bg_seqset_find_rows.argtypes = [vp, vp, vp, u64, i32, u64, vp]
    L.bg_readmap_rows.argtypes = [vp, vp, i32, vp, u32, vp, vp, u64, P(u64)]
    L.bg_readmap_shard_build.argtypes = [vp, vp, u64, P(vp)]
    L.bg_readmap_shard_size.argtypes = [vp]
    L.bg_readmap_shard_size.restype = u64
    L.bg_readmap_shard_free.argtypes = [vp]
    L.bg_readmap_shard_free.restype = None
    L.bg_readmap_shard_first_claims.argtypes = [vp, vp, u64, u32, vp, vp, u64, P(u64)]
    L.bg_readmap_shard_apply_claims.argtypes = [vp, vp, u64, i32, vp, u64, u32, vp, vp, vp, P(u64)]
    L.bg_readmap_shard_take_replies.argtypes = [vp, vp, u64, vp, vp, u64]
    L.bg_readmap_shard_columns.argtypes = [vp, vp, vp, vp, vp, vp]
    L.bg_prefix_reads_batch.argtypes = [vp, vp, vp, u64, vp, vp, vp, i32, u32, vp, vp]
    L.bg_read_info_batch.argtypes = [vp, vp, vp, u64, vp, vp, vp, vp, vp, vp, vp]
    L.bg_propagate_from_end_batch.argtypes = [vp, vp, vp, u64, vp, vp, vp, vp, vp, vp, u32, vp, vp, vp, vp, vp, vp,
                                              vp, vp, vp, vp, u32, vp, vp, vp, vp]
    L.bg_push_to_pair_trace_batch.argtypes = [vp, vp, u64, vp, vp, vp, vp, u32, u32, u32, u32] + [vp] * 8
    L.bg_seqset_merge.argtypes = [vp, u32, P(vp), P(vp), P(vp)]
    L.bg_readmap_migrate.argtypes = [vp, vp, vp, vp, P(vp)]
    i64 = C.c_int64
    L.bg_asm_list_create.argtypes = [P(vp)]
    L.bg_asm_list_free.argtypes = [vp]
    L.bg_asm_list_free.restype = None
    L.bg_asm_list_size.argtypes = [vp]
    L.bg_asm_list_size.restype = u64
    L.bg_asm_list_add.argtypes = [vp, i64, i64, C.c_char_p, u64, i32, C.c_char_p]
    L.bg_asm_list_sort.argtypes = [vp]
    L.bg_asm_list_get.argtypes = [vp, u64, P(i64), P(i64), P(C.c_char_p), P(u64), P(i32), P(C.c_char_p)]
    L.bg_asm_list_rc_path.argtypes = [vp, u64, P(u64), P(u64), vp, vp, vp]
    L.bg_make_ref_assemblies.argtypes = [vp, u64, P(i64), P(C.c_char_p), P(u64), i64, i64, i64]
    L.bg_graph_trim_ref.argtypes = [vp, u64, P(i64), P(C.c_char_p), P(u64), i64]
    L.bg_dedup_assemblies.argtypes = [vp]
    L.bg_update_rc_seqset_entries.argtypes = [vp, vp, vp, vp, u32]
    L.bg_discover_branch.argtypes = [vp, vp, vp, vp, u32, C.c_char_p, C.c_char_p]
    L.bg_discover_push_to_pair.argtypes = [vp, vp, vp, vp, u32, i64, C.c_char_p, C.c_char_p]
    L.bg_create_sample.argtypes = [vp, vp, vp, u64, P(CreateParams), i32, P(vp), P(vp), P(BuildStats)]
    L.bg_partition_rows.argtypes = [vp, vp, u64, i32, i32, u32, vp, vp, vp]
    L.bg_ctx_memory.argtypes = [vp, P(u64), P(u64), P(u64), i32]
    L.bg_bwt_open.argtypes = [C.c_char_p, P(vp)]
    L.bg_bwt_free.argtypes = [vp]
    L.bg_bwt_free.restype = None
    L.bg_bwt_entries.argtypes = [vp]
    L.bg_bwt_entries.restype = u64
    L.bg_bwt_find.argtypes = [vp, C.c_char_p, u64, P(u64), P(u64)]
    L.bg_bwt_match.argtypes = [vp, u64, P(u64)]
    L.bg_reference_open.argtypes = [C.c_char_p, P(vp)]
    L.bg_reference_free.argtypes = [vp]
    L.bg_reference_free.restype = None
    L.bg_reference_size.argtypes = [vp]
    L.bg_reference_size.restype = u64
    L.bg_reference_scaffold_count.argtypes = [vp]
    L.bg_reference_scaffold_count.restype = u32
    L.bg_reference_scaffold_name.argtypes = [vp, u32]
    L.bg_reference_scaffold_name.restype = C.c_char_p
    L.bg_reference_scaffold_len.argtypes = [vp, u32]
    L.bg_reference_scaffold_len.restype = u64
    L.bg_reference_supercontig_count.argtypes = [vp]
    L.bg_reference_supercontig_count.restype = u32
    L.bg_reference_supercontig.argtypes = [vp, u32, P(u32), P(u64), P(u64), P(u64)]
    L.bg_reference_bases.argtypes = [vp, u64, u64, C.c_char_p]
    L.bg_kmer_instances.argtypes = [vp, vp, i32, P(u64)]
    L.bg_kmer_extract.argtypes = [vp, vp, i32, vp, u64]
    L.bg_kmer_extract_by_owner.argtypes = [vp, vp, i32, u32, vp, u64, vp]
    L.bg_kmer_count_keys.argtypes = [vp, vp, u64, i32, u32, P(vp)]
    L.bg_skm_supported.argtypes = [i32]
    L.bg_skm_capacity.argtypes = [vp, vp, i32, u64, P(u64)]
    L.bg_skm_extract_by_owner.argtypes = [vp, vp, i32, u64, u32, vp, vp, u64, vp, P(u64)]
    L.bg_skm_count.argtypes = [vp, vp, u64, i32, u32, P(vp), vp]
    L.bg_kmerset_attach_presence.argtypes = [vp, vp, vp, vp, vp, vp, u64]
    L.bg_kmerset_export.argtypes = [vp, vp, vp]
    L.bg_kmerset_from_kmers.argtypes = [vp, vp, u64, i32, vp, P(vp)]
    L.bg_reads_words.argtypes = [vp]
    L.bg_reads_export.argtypes = [vp, vp, i32, vp, vp]
    L.bg_reads_import.argtypes = [vp, u64, i32, u32, vp, vp, P(vp)]
    L.bg_suffix_records.argtypes = [vp, vp, u64, u64, i32, P(vp)]
    L.bg_records_prefix_hist.argtypes = [vp, vp, i32, u64, vp]
    L.bg_records_count.restype = u64
    L.bg_records_count.argtypes = [vp]
    L.bg_records_words.argtypes = [vp]
    L.bg_records_rows.argtypes = [vp, vp, vp]
    L.bg_records_from_rows.argtypes = [vp, vp, u64, i32, P(vp)]
    L.bg_records_free.argtypes = [vp]
    L.bg_records_free.restype = None
    L.bg_entries_from_records.argtypes = [vp, vp, P(vp), P(BuildStats)]
    L.bg_seqset_words.argtypes = [vp]
    L.bg_seqset_boundary_rows.argtypes = [vp, vp, vp, vp]
    L.bg_seqset_fix_boundary.argtypes = [vp, vp, vp, vp]
    L.bg_seqset_tail_rows.argtypes = [vp, vp, vp]
    L.bg_seqset_entry_rows.argtypes = [vp, vp, vp]
    L.bg_seqset_apply_tails.argtypes = [vp, vp, vp, u64]
    L.bg_seqset_slice_export.argtypes = [vp, vp, vp, vp, vp]
    L.bg_seqset_assemble.argtypes = [vp, u32, vp, vp, vp, vp, u32, P(vp)]
    L.bg_seqset_slice_masks.argtypes = [vp, vp, vp, vp, vp]
    L.bg_seqset_slice_build.argtypes = [vp, u64, vp, vp, vp, u32, P(vp)]
    L.bg_seqset_slice_popcounts.argtypes = [vp, P(u64)]
    L.bg_seqset_sharded_create_file.argtypes = [C.c_char_p, u64, u32, P(u64), i32, P(C.c_char_p), P(u64), vp]
    L.bg_seqset_slice_write.argtypes = [vp, vp, C.c_char_p, P(u64), u64, i32, P(u64)]
    L.bg_readmap_has_mate_loop.argtypes = [vp]
    L.bg_rows_select.argtypes = [vp, vp, u64, i32, vp, vp, u64, P(u64)]
    L.bg_seqset_uuid.argtypes = [vp, C.c_char_p, u64]
    L.bg_seqset_tail_rows2.argtypes = [vp, vp, u32, vp, vp]
    L.bg_seqset_apply_tails2.argtypes = [vp, vp, vp, u64, vp]
    L.bg_seqset_tail_rows_full.argtypes = [vp, vp, vp, vp, u64, vp, u64, P(u64)]
    L.bg_records_from_row_parts.argtypes = [vp, u32, P(vp), P(u64), i32, P(vp)]
    L.bg_fastq_import.argtypes = [vp, u32, P(vp), P(u64), i32, i32, u32, u32, P(vp), P(vp), P(u64), P(u64)]
    L.bg_dev_upload_pinned.argtypes = [vp, vp, vp, u64, i32]
    L.bg_dev_upload_file.argtypes = [vp, vp, i32, u64, u64, i32]
    L.bg_oriented_rows.argtypes = [vp, vp, i32, vp]
    L.bg_rows_prefix_hist.argtypes = [vp, vp, u64, i32, i32, u64, vp]
    L.bg_cover_build.argtypes = [vp, vp, u64, i32, i32, P(vp)]
    L.bg_cover_test.argtypes = [vp, vp, vp, u64, vp]
    L.bg_cover_free.argtypes = [vp]
    L.bg_cover_free.restype = None
    L.bg_cover_query_begin.argtypes = [vp, vp, u64, i32, i32, P(vp)]
    L.bg_cover_query_pending.argtypes = [vp]
    L.bg_cover_query_pending.restype = u64
    L.bg_cover_query_candidates.argtypes = [vp, vp, i32, i32, vp, u64, P(u64)]
    L.bg_cover_query_candidates_split.argtypes = [vp, vp, i32, i32, u32, vp, vp, u64, vp, P(u64)]
    L.bg_cover_query_answers.argtypes = [vp, vp, vp, vp, u64, vp]
    L.bg_cover_query_plain.argtypes = [vp, vp, vp, u64, P(u64)]
    L.bg_cover_query_free.argtypes = [vp]
    L.bg_cover_query_free.restype = None
    L.bg_kmerset_build_hash.argtypes = [vp, vp]
    L.bg_skm_lookup.argtypes = [vp, vp, vp, u64, vp, i32]
    L.bg_corr_begin.argtypes = [vp, vp, i32, P(CorrectParams), vp, vp, vp, u64, P(vp)]
    L.bg_corr_step.argtypes = [vp, vp, P(u64), P(vp)]
    L.bg_corr_answers.argtypes = [vp, vp, vp, vp, u64]
    L.bg_corr_finish.argtypes = [vp, vp, P(vp), P(CorrectStats)]
    L.bg_corr_probes.argtypes = [vp]
    L.bg_corr_probes.restype = u64
    L.bg_corr_free.argtypes = [vp]
    L.bg_corr_free.restype = None
    _lib = L
    return L


def check(rc):
    if rc != BG_OK:
        msg = lib().bg_last_error()
        raise BgError(rc, msg.decode() if msg else "")


def np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)
