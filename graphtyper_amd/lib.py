"""ctypes binding of libgtx.so (C ABI in include/gtx.h) + helpers that build the SoA graph view from variant records.

PyTorch is used by callers only for device memory, streams and torch.distributed; everything that touches reads runs
inside libgtx's HIP kernels.  There is no CPU path: without the built library or without a GPU the calls fail loudly.
"""
import ctypes as C
import os
import sys
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, os.environ.get("GTX_LIB", "libgtx.so"))  # GTX_LIB=libgtx_prof.so selects the phase-timing build
INVALID_ID = 0xFFFFFFFF
SPECIAL_START = 0xD0000000

ST_LABEL_OVERFLOW, ST_PATH_OVERFLOW, ST_DFS_OVERFLOW, ST_RECORD_OVERFLOW, ST_EXTERNAL = 1, 2, 4, 8, 16
ST_ERROR_MASK = 15

# names every build of libgtx.so must export (checked by tests/test_abi.py against include/gtx.h)
EXPORTS = ["gtx_strerror", "gtx_last_error", "gtx_ctx_create", "gtx_ctx_destroy", "gtx_ctx_special_positions",
           "gtx_ctx_score_layout", "gtx_ctx_haplotypes", "gtx_ctx_near_pairs", "gtx_index_stats", "gtx_index_get", "gtx_index_dump", "gtx_ctx_hint_table",
           "gtx_align_batch", "gtx_score_batch", "gtx_calls_batch", "gtx_ctx_big_records", "gtx_ctx_big_records_rewind", "gtx_ctx_exact_pass_tasks", "gtx_graph_sv_table", "gtx_ctx_pass_times", "gtx_ctx_error_count", "gtx_ctx_profile", "gtx_ctx_profile_log", "gtx_records_failed", "gtx_vcf_sites", "gtx_vcf_records_final", "gtx_align_batch_planes_compact", "gtx_score_batch_compact", "gtx_align_batch_planes_triaged", "gtx_score_batch_queued", "gtx_scores_replay_compact", "gtx_scores_replay_log", "gtx_scores_replay_apply", "gtx_scores_finalize", "gtx_phase_flags", "gtx_stream_create",
           "gtx_stream_destroy", "gtx_stream_push", "gtx_stream_set_coverage", "gtx_stream_finish", "gtx_stream_counts", "gtx_graph_build", "gtx_graph_from_files", "gtx_graph_get_view",
           "gtx_graph_destroy",
           "gtx_scores_alloc", "gtx_scores_zero", "gtx_scores_free", "gtx_scores_reduce", "gtx_comm_unique_id", "gtx_comm_init_rank",
           "gtx_comm_destroy", "gtx_ctx_kernel_times", "gtx_ref_depth_finalize", "gtx_vcf_records", "gtx_scores_replay", "gtx_reads_open", "gtx_reads_info",
           "gtx_reads_sample_name", "gtx_reads_next", "gtx_reads_close", "gtx_align_batch_flags", "gtx_score_batch_flags", "gtx_score_batch_words", "gtx_item_words",
           "gtx_pack_planes", "gtx_reads_to_planes", "gtx_align_batch_planes", "gtx_align_batch_planes_staged", "gtx_stream_set_planes", "gtx_device_cache_release",
           "gtx_disc_create", "gtx_disc_destroy", "gtx_disc_events_batch", "gtx_disc_first_pass", "gtx_vcf_header", "gtx_bgzf_compress",
           "gtx_shrink_params_default", "gtx_bam_shrink", "gtx_inflate_raw", "gtx_tabix_build", "gtx_tabix_start", "gtx_pipeline_run", "gtx_regions_run", "gtx_regions_free", "gtx_bam_shrink_multi", "gtx_disc_first_pass_haplotypes", "gtx_disc_merge"]


class GraphView(C.Structure):
    _fields_ = [("n_ref", C.c_uint32), ("n_var", C.c_uint32),
                ("ref_order", C.c_void_p), ("ref_len", C.c_void_p), ("ref_dna_off", C.c_void_p), ("ref_nvar", C.c_void_p),
                ("ref_first_var", C.c_void_p), ("var_order", C.c_void_p), ("var_len", C.c_void_p),
                ("var_dna_off", C.c_void_p), ("var_out_ref", C.c_void_p), ("dna", C.c_void_p), ("dna_len", C.c_uint64),
                ("event_off", C.c_void_p), ("event_val", C.c_void_p)]


class PipelineStats(C.Structure):
    _fields_ = [("records", C.c_uint64), ("tasks", C.c_uint64), ("items", C.c_uint64), ("decode_s", C.c_double), ("push_s", C.c_double),
                ("enqueue_s", C.c_double), ("slowest_thread_s", C.c_double), ("loop_s", C.c_double), ("wall_s", C.c_double),
                ("n_samples", C.c_uint32), ("n_threads", C.c_uint32), ("records_failed", C.c_uint64), ("score_items_refused", C.c_uint64),
                ("connections_dropped", C.c_uint64)]


class RegionsStats(C.Structure):
    _fields_ = [("graph_build_s", C.c_double), ("ctx_create_s", C.c_double), ("device_s", C.c_double), ("vcf_text_s", C.c_double), ("wall_s", C.c_double),
                ("n_builders", C.c_uint32), ("n_device_threads", C.c_uint32), ("n_text_threads", C.c_uint32), ("reserved", C.c_uint32),
                ("records_failed", C.c_uint64), ("score_items_refused", C.c_uint64), ("connections_dropped", C.c_uint64)]


class ShrinkParams(C.Structure):
    _fields_ = [("max_frag_len", C.c_int32), ("min_num_matching", C.c_int32), ("filter_mapq0", C.c_int32), ("no_filter_on_coverage", C.c_int32),
                ("min_read_len", C.c_int32), ("min_read_len_low_mapq", C.c_int32), ("min_unpaired_read_len", C.c_int32), ("sam_flag_filter", C.c_int32),
                ("as_filter_threshold", C.c_int64), ("avg_cov_by_readlen", C.c_double), ("change_read_names", C.c_int32), ("compress_level", C.c_int32)]


class ShrinkStats(C.Structure):
    _fields_ = [("records_read", C.c_uint64), ("records_written", C.c_uint64), ("pairs_kept", C.c_uint64), ("singles_kept", C.c_uint64),
                ("dropped_by_depth", C.c_uint64)]


class Params(C.Structure):
    _fields_ = [("max_index_labels", C.c_int32), ("is_sv_graph", C.c_int32), ("hq_reads", C.c_int32),
                ("force_align_both_orientations", C.c_int32), ("is_segment_calling", C.c_int32),
                ("sam_flag_filter", C.c_int32), ("no_second_pass", C.c_int32), ("big_record_words", C.c_uint32),
                ("exact_pass_mb", C.c_uint32)]


class ScoreLayout(C.Structure):
    _fields_ = [("n_hap", C.c_uint32), ("total_tri", C.c_uint64), ("total_allele", C.c_uint64), ("total_near", C.c_uint64),
                ("ref_depth_len", C.c_uint32)]


class ScoreBuffers(C.Structure):
    _fields_ = [("n_samples", C.c_uint32), ("d_log_score", C.c_void_p), ("d_gt_cov", C.c_void_p), ("d_hap_u32", C.c_void_p),
                ("d_stat_u64", C.c_void_p), ("d_stat_u32", C.c_void_p), ("d_conn_log", C.c_void_p),
                ("d_conn_count", C.c_void_p), ("conn_cap", C.c_uint32), ("d_conn_near", C.c_void_p),
                ("d_ref_depth", C.c_void_p), ("ref_depth_len", C.c_uint32)]


READ_META = np.dtype([("l_qseq", np.uint16), ("flag", np.uint16), ("tid", np.int32), ("mtid", np.int32), ("isize", np.int32), ("pos", np.int32)],
                     align=True)
REC_META = np.dtype([("align_index", np.uint32), ("flag", np.uint16), ("mapq", np.uint8), ("score_diff", np.uint8),
                     ("pos", np.int32), ("isize", np.int32)], align=True)
SCORE_ITEM = np.dtype([("first", REC_META), ("second", REC_META), ("sample", np.uint32), ("kind", np.uint32)], align=True)
ITEM_LEFTOVER = 1
FLAG_FORWARD_ONLY = 0x8000
SAMPLE_CALL = np.dtype([("gt_first", np.uint16), ("gt_second", np.uint16), ("ref_total_depth", np.uint16),
                        ("alt_total_depth", np.uint16), ("gq", np.uint8), ("ambiguous_depth", np.uint8),
                        ("alt_proper_pair_depth", np.uint8), ("reserved", np.uint8)], align=True)
PHASE_ENTRY = np.dtype([("hap1", np.uint16), ("allele1", np.uint16), ("hap2", np.uint16), ("allele2", np.uint16),
                        ("flags", np.int8), ("reserved", np.uint8)], align=True)
STREAM_RECORD = np.dtype([("flag", np.uint16), ("mapq", np.uint8), ("score_diff", np.uint8), ("tid", np.int32),
                          ("mtid", np.int32), ("pos", np.int32), ("isize", np.int32), ("l_qseq", np.uint16),
                          ("rg", np.uint16), ("sample", np.uint32), ("name_id", np.uint64), ("mpos", np.int32),
                          ("n_cigar", np.uint32), ("cigar_front", np.uint32), ("cigar_back", np.uint32)], align=True)
DISC_READ = np.dtype([("pos", np.int32), ("flag", np.uint16), ("mapq", np.uint8), ("reserved", np.uint8), ("l_qseq", np.uint16),
                      ("n_cigar", np.uint16), ("cigar_off", np.uint32)], align=True)
DISC_EVENT = np.dtype([("read", np.uint32), ("pos", np.uint32), ("seq", np.uint32), ("len", np.uint16), ("type", np.uint8), ("hq", np.uint8),
                       ("max_distance", np.uint16), ("reserved", np.uint16)], align=True)
DISC_READ_OUT = np.dtype([("first_event", np.uint32), ("n_events", np.uint32), ("pos_end", np.int32), ("state", np.uint32)], align=True)
assert DISC_READ.itemsize == 16 and DISC_EVENT.itemsize == 20 and DISC_READ_OUT.itemsize == 16
LABEL = np.dtype([("start_index", np.uint32), ("end_index", np.uint32), ("variant_id", np.uint32)], align=True)
assert READ_META.itemsize == 20 and REC_META.itemsize == 16 and SCORE_ITEM.itemsize == 40 and STREAM_RECORD.itemsize == 56 \
    and SAMPLE_CALL.itemsize == 12


def build(force=False):
    """compile libgtx.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)"""
    src_dir = os.path.join(HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if not f.startswith("build")] + [os.path.join(ROOT, "include", "gtx.h")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", src_dir, "-s", "-j", str(min(8, os.cpu_count() or 1))])
    return LIB_PATH


_lib = None


def lib():
    """the loaded library; raises if it has not been built (never falls back to anything else)"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libgtx.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        # A process that uses PyTorch as well has to have PyTorch's HIP runtime in it FIRST: libgtx.so names libamdhip64 by its
        # soname, and with torch already loaded that resolves to the runtime torch brought -- one runtime for both.  The other way
        # round the process holds two, and the library's own sees no device ("no HIP device visible").  Whoever binds the library
        # from Python gets the order right here; a C or C++ host has one runtime anyway.  GTX_NO_TORCH=1: a process that will never
        # use PyTorch (and does not want its two seconds of import) loads the library alone.
        if "torch" not in sys.modules and not os.environ.get("GTX_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        L.gtx_strerror.restype = C.c_char_p
        L.gtx_strerror.argtypes = [C.c_int]
        L.gtx_last_error.restype = C.c_char_p
        L.gtx_ctx_create.argtypes = [C.POINTER(GraphView), C.POINTER(Params), C.c_int, C.POINTER(C.c_void_p)]
        L.gtx_ctx_destroy.argtypes = [C.c_void_p]
        L.gtx_ctx_special_positions.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p, C.c_uint32]
        L.gtx_ctx_score_layout.argtypes = [C.c_void_p, C.POINTER(ScoreLayout)]
        L.gtx_ctx_haplotypes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gtx_index_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gtx_index_get.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.gtx_index_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gtx_ctx_hint_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.gtx_align_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                      C.c_void_p]
        L.gtx_score_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(ScoreBuffers),
                                      C.c_void_p]
        L.gtx_calls_batch.argtypes = [C.c_void_p, C.POINTER(ScoreBuffers), C.c_void_p, C.c_void_p, C.c_void_p]
        L.gtx_ctx_big_records.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_uint64)]
        L.gtx_ctx_big_records_rewind.argtypes = [C.c_void_p, C.c_void_p]
        L.gtx_ctx_exact_pass_tasks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.gtx_graph_sv_table.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.gtx_ctx_pass_times.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        L.gtx_ctx_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        L.gtx_ctx_error_count.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.gtx_ctx_profile.argtypes = [C.c_void_p, C.c_void_p]
        L.gtx_align_batch_planes_compact.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gtx_score_batch_compact.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                              C.POINTER(ScoreBuffers), C.c_void_p]
        L.gtx_align_batch_planes_triaged.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gtx_score_batch_queued.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.POINTER(ScoreBuffers), C.c_void_p]
        L.gtx_scores_replay_compact.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                                C.POINTER(ScoreBuffers), C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gtx_scores_replay_log.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(ScoreBuffers),
                                            C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gtx_scores_replay_apply.argtypes = [C.c_void_p, C.POINTER(ScoreBuffers), C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
        L.gtx_records_failed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
        L.gtx_ctx_profile_log.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.gtx_scores_finalize.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                          C.POINTER(C.c_uint64)]
        L.gtx_phase_flags.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64,
                                      C.POINTER(C.c_uint64)]
        L.gtx_ctx_near_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.gtx_ref_depth_finalize.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
        L.gtx_vcf_records.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.gtx_vcf_records_final.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.gtx_vcf_sites.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.gtx_align_batch_flags.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.gtx_disc_create.argtypes = [C.c_char_p, C.c_uint64, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
        L.gtx_disc_destroy.argtypes = [C.c_void_p]
        L.gtx_disc_destroy.restype = None
        L.gtx_disc_events_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                            C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gtx_disc_first_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32,
                                          C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.gtx_disc_first_pass_haplotypes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32,
                                                     C.c_uint32, C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.gtx_disc_merge.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.gtx_vcf_header.argtypes = [C.POINTER(VcfHeaderRequest), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.gtx_bgzf_compress.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.gtx_device_cache_release.argtypes = []
        L.gtx_device_cache_release.restype = None
        L.gtx_pack_planes.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.gtx_reads_to_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.gtx_score_batch_words.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(ScoreBuffers), C.c_void_p]
        L.gtx_item_words.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.gtx_align_batch_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.gtx_align_batch_planes_staged.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p]
        L.gtx_stream_set_planes.argtypes = [C.c_void_p, C.c_uint32]
        L.gtx_score_batch_flags.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(ScoreBuffers), C.c_void_p]
        L.gtx_reads_open.argtypes = [C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p, C.POINTER(C.c_void_p)]
        L.gtx_reads_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.gtx_reads_sample_name.argtypes = [C.c_void_p, C.c_uint32]
        L.gtx_reads_sample_name.restype = C.c_char_p
        L.gtx_reads_next.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        L.gtx_reads_close.argtypes = [C.c_void_p]
        L.gtx_reads_close.restype = None
        L.gtx_bam_shrink_multi.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(ShrinkParams), C.c_char_p, C.POINTER(ShrinkStats)]
        L.gtx_inflate_raw.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.gtx_regions_run.argtypes = [C.POINTER(RegionJob), C.c_uint32, C.POINTER(Params), C.c_int, C.c_char_p, C.POINTER(C.c_char_p), C.c_uint32, C.c_uint32,
                                      C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(RegionsStats)]
        L.gtx_regions_free.argtypes = [C.POINTER(RegionJob), C.c_uint32]
        L.gtx_regions_free.restype = None
        L.gtx_pipeline_run.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint64,
                                       C.POINTER(ScoreBuffers), C.POINTER(PipelineStats)]
        L.gtx_tabix_build.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
        L.gtx_tabix_start.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.gtx_shrink_params_default.argtypes = [C.POINTER(ShrinkParams)]
        L.gtx_shrink_params_default.restype = None
        L.gtx_bam_shrink.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(ShrinkParams),
                                     C.c_char_p, C.POINTER(ShrinkStats)]
        L.gtx_scores_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gtx_stream_create.argtypes = [C.POINTER(Params), C.c_uint32, C.POINTER(C.c_void_p)]
        L.gtx_stream_destroy.argtypes = [C.c_void_p]
        L.gtx_stream_push.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                      C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.gtx_stream_set_coverage.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.gtx_stream_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.gtx_stream_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gtx_graph_build.argtypes = [C.c_char_p, C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_void_p)]
        L.gtx_graph_from_files.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.gtx_graph_get_view.argtypes = [C.c_void_p, C.POINTER(GraphView)]
        L.gtx_graph_destroy.argtypes = [C.c_void_p]
        L.gtx_scores_alloc.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(ScoreBuffers), C.POINTER(C.c_uint64)]
        L.gtx_scores_zero.argtypes = [C.c_void_p, C.POINTER(ScoreBuffers), C.c_void_p]
        L.gtx_scores_free.argtypes = [C.c_void_p, C.POINTER(ScoreBuffers)]
        L.gtx_scores_reduce.argtypes = [C.c_void_p, C.POINTER(ScoreBuffers), C.c_void_p, C.c_void_p]
        L.gtx_comm_unique_id.argtypes = [C.c_void_p]
        L.gtx_comm_init_rank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.gtx_comm_destroy.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class GtxError(RuntimeError):
    def __init__(self, status):
        L = lib()
        self.status = status
        super().__init__("%s: %s" % (L.gtx_strerror(status).decode(), L.gtx_last_error().decode()))


def check(status):
    if status != 0:
        raise GtxError(status)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


_hip = None


def download(ptr, dtype, count):
    """host copy of `count` elements of `dtype` at the device pointer `ptr` (hipMemcpy, synchronous)"""
    global _hip
    out = np.zeros(int(count), dtype)
    if out.nbytes:
        if _hip is None:
            _hip = C.CDLL("libamdhip64.so")
            _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        rc = _hip.hipMemcpy(_p(out), C.c_void_p(int(ptr)), C.c_size_t(out.nbytes), C.c_int(2))  # hipMemcpyDeviceToHost
        if rc != 0:
            raise RuntimeError("hipMemcpy failed (%d)" % rc)
    return out


# ------------------------------------------------------------------------------------------------------------------
# binding of the host graph builder gtx_graph_build (Graph::add_genomic_region, src/graph/graph.cpp:41-339, including the
# record merge rules) -- SURVEY.md 8(f) row 1
# ------------------------------------------------------------------------------------------------------------------
def parse_events(info):
    """GT_ID / GT_ANTI_HAPLOTYPE of a single-alt record (src/graph/constructor.cpp:1540-1588) ->
    (ref_events, alt_events, alt_anti_events)"""
    ref_ev, alt_ev, alt_anti = [], [], []
    if info and info != ".":
        for kv in info.split(";"):
            if "=" not in kv:
                continue
            k, v = kv.split("=", 1)
            if k == "GT_ID":
                ref_ev.append(-int(v))
                alt_ev.append(int(v))
            elif k == "GT_ANTI_HAPLOTYPE":
                alt_anti.extend(int(x) for x in v.split(","))
    return ref_ev, alt_ev, alt_anti


class _Allele(C.Structure):
    _fields_ = [("seq", C.c_char_p), ("len", C.c_uint32), ("events", C.POINTER(C.c_int64)), ("n_events", C.c_uint32),
                ("anti_events", C.POINTER(C.c_int64)), ("n_anti_events", C.c_uint32)]


class _Record(C.Structure):
    _fields_ = [("pos", C.c_uint32), ("n_alleles", C.c_uint32), ("alleles", C.POINTER(_Allele)), ("is_sv", C.c_int32)]


class RegionJob(C.Structure):
    _fields_ = [("reference", C.c_char_p), ("reference_len", C.c_uint64), ("region_begin", C.c_int64), ("region_end", C.c_int64),
                ("records", C.POINTER(_Record)), ("n_records", C.c_uint32), ("add_all_variants", C.c_int32), ("d_planes", C.c_void_p),
                ("plane_stride", C.c_uint32), ("d_meta", C.c_void_p), ("n_reads", C.c_uint64), ("d_items", C.c_void_p), ("n_items", C.c_uint64),
                ("vcf_begin", C.c_uint32), ("vcf_end", C.c_uint32), ("filter_zero_qual", C.c_int32), ("status", C.c_int32),
                ("text", C.c_void_p), ("text_len", C.c_uint64)]


def _explicit_events(info, n_alts):
    """test syntax used by tests/test_graph_vectors.py: RE= / RA= (reference allele events / anti events),
    E<i>= / A<i>= (alt i); comma separated integers"""
    ev = [([], []) for _ in range(n_alts + 1)]
    if info and info != ".":
        for kv in info.split(";"):
            if "=" not in kv:
                continue
            k, v = kv.split("=", 1)
            if k in ("RE", "RA") or (k[0] in "EA" and k[1:].isdigit()):
                vals = [int(x) for x in v.split(",") if x]
                if k == "RE":
                    ev[0][0].extend(vals)
                elif k == "RA":
                    ev[0][1].extend(vals)
                elif int(k[1:]) < n_alts:
                    ev[int(k[1:]) + 1][0 if k[0] == "E" else 1].extend(vals)
    return ev


def records_array(records):
    """[(pos0, ref, [alts], info)] -> (array of gtx_record, what it points into: keep both alive while the array is in use)"""
    keep = []
    recs = (_Record * max(len(records), 1))()
    for i, (pos, ref, alts, info) in enumerate(records):
        ev = _explicit_events(info, len(alts))
        if len(alts) == 1:
            r_ev, a_ev, a_anti = parse_events(info)
            ev[0][0].extend(r_ev)
            ev[1][0].extend(a_ev)
            ev[1][1].extend(a_anti)
        als = (_Allele * (len(alts) + 1))()
        for j, seq in enumerate([ref] + list(alts)):
            b = seq.encode()
            e = (C.c_int64 * max(len(ev[j][0]), 1))(*ev[j][0])
            a = (C.c_int64 * max(len(ev[j][1]), 1))(*ev[j][1])
            keep.extend([b, e, a])
            als[j] = _Allele(b, len(b), e, len(ev[j][0]), a, len(ev[j][1]))
        keep.append(als)
        recs[i] = _Record(pos, len(alts) + 1, als, 0)
    return recs, keep


def graph_from_records(reference, records, region_begin=0, region_end=0xFFFFFFFF, add_all_variants=False,
                       extend_prefix=False, is_sv_graph=False):
    """reference: str of the region; records: [(pos0, ref, [alts], info)] sorted by pos0 (contig coordinates, 0-based).
    Calls gtx_graph_build (C++: filters, record merging, node emission) and returns the node tables as numpy arrays laid
    out as gtx_graph_view expects."""
    L = lib()
    recs, keep = records_array(records)
    h = C.c_void_p()
    refb = reference.encode()
    check(L.gtx_graph_build(refb, len(refb), region_begin, region_end, recs, len(records), int(add_all_variants),
                            int(is_sv_graph), int(extend_prefix), C.byref(h)))
    return _graph_tables(h)


def _graph_tables(h):
    """node tables of a gtx_graph as numpy arrays laid out as gtx_graph_view expects; destroys the handle"""
    L = lib()
    try:
        v = GraphView()
        check(L.gtx_graph_get_view(h, C.byref(v)))

        def arr(ptr, n, dt):
            if not ptr or n == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,)).copy()

        R, V = v.n_ref, v.n_var
        g = dict(ref_order=arr(v.ref_order, R, np.uint32), ref_len=arr(v.ref_len, R, np.uint32),
                 ref_dna_off=arr(v.ref_dna_off, R, np.uint32), ref_nvar=arr(v.ref_nvar, R, np.uint32),
                 ref_first_var=arr(v.ref_first_var, R, np.uint32), var_order=arr(v.var_order, V, np.uint32),
                 var_len=arr(v.var_len, V, np.uint32), var_dna_off=arr(v.var_dna_off, V, np.uint32),
                 var_out_ref=arr(v.var_out_ref, V, np.uint32), dna=arr(v.dna, v.dna_len, np.uint8))
        if v.event_off:
            g["event_off"] = arr(v.event_off, 2 * V + 1, np.uint32)
            g["event_val"] = arr(v.event_val, int(g["event_off"][-1]), np.int64)
        else:
            g["event_off"] = np.zeros(2 * V + 1, np.uint32)
            g["event_val"] = np.zeros(0, np.int64)
        return g
    finally:
        L.gtx_graph_destroy(h)


def graph_from_files(fasta, vcf, region, add_all_variants=False, is_sv_graph=False, with_sv_table=False):
    """gtx_graph_from_files (construct_graph, src/graph/constructor.cpp:1597-1777): node tables + (begin, end) of the
    reference span that was read (0-based) [+ Graph::SVs as text: gtx_graph_sv_table]"""
    L = lib()
    h = C.c_void_p()
    b, e = C.c_int64(), C.c_int64()
    check(L.gtx_graph_from_files(str(fasta).encode(), (str(vcf) if vcf else "").encode(), region.encode(), int(add_all_variants),
                                 int(is_sv_graph), C.byref(h), C.byref(b), C.byref(e)))
    table = None
    if with_sv_table:
        n = C.c_uint64()
        check(L.gtx_graph_sv_table(h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(int(n.value) + 1)
        check(L.gtx_graph_sv_table(h, buf, int(n.value), C.byref(n)))
        table = buf.raw[:int(n.value)].decode()
    tables = _graph_tables(h)
    return (tables, (int(b.value), int(e.value)), table) if with_sv_table else (tables, (int(b.value), int(e.value)))


def pack_nibbles(codes, stride=None):
    """[n, L] 4-bit codes -> [n, stride] BAM-packed bytes (high nibble first)"""
    codes = np.asarray(codes, np.uint8)
    n, L = codes.shape
    nb = (L + 1) // 2
    stride = stride or ((nb + 15) // 16) * 16
    if L % 2:
        codes = np.concatenate([codes, np.zeros((n, 1), np.uint8)], axis=1)
    out = np.zeros((n, stride), np.uint8)
    out[:, :nb] = (codes[:, 0::2] << 4) | codes[:, 1::2]
    return out


def pack_planes(seq, plane_stride=None):
    """gtx_pack_planes: [n, stride] BAM nibble rows -> [n, plane_stride] plane rows (include/gtx.h: reads as bit planes)"""
    seq = np.ascontiguousarray(seq, np.uint8)
    n, stride = seq.shape
    plane_stride = plane_stride or ((stride + 15) // 16) * 16
    out = np.zeros((n, plane_stride), np.uint8)
    check(lib().gtx_pack_planes(_p(seq), stride, n, _p(out), plane_stride))
    return out


def item_words(items):
    """gtx_item_words: the compact form of score items for gtx_score_batch_words (one uint32 per item)"""
    items = np.ascontiguousarray(items, SCORE_ITEM)
    out = np.zeros(len(items), np.uint32)
    check(lib().gtx_item_words(_p(items), len(items), _p(out)))
    return out


def planes_reference(codes, plane_stride):
    """the plane layout restated in numpy (tests): [n, L] codes -> [n, plane_stride] bytes"""
    codes = np.asarray(codes, np.uint8)
    n, L = codes.shape
    groups = plane_stride // 16
    padded = np.zeros((n, groups * 32), np.uint8)
    padded[:, :L] = codes
    out = np.zeros((n, groups, 4), np.uint32)
    for b in range(4):
        bits = ((padded >> b) & 1).reshape(n, groups, 32).astype(np.uint32)
        out[:, :, b] = (bits << np.arange(32, dtype=np.uint32)).sum(axis=2, dtype=np.uint64).astype(np.uint32)
    return out.reshape(n, groups * 4).view(np.uint8).reshape(n, plane_stride)


class Reads:
    """gtx_reads: BAM files merged into the record stream gtx_stream_push takes"""

    def __init__(self, paths, region=None):
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        h = C.c_void_p()
        check(lib().gtx_reads_open(arr, len(paths), region.encode() if region else None, C.byref(h)))
        self.h = h
        ns, nrg = C.c_uint32(), C.c_uint32()
        check(lib().gtx_reads_info(self.h, C.byref(ns), C.byref(nrg)))
        self.n_read_groups = int(nrg.value)
        self.samples = [lib().gtx_reads_sample_name(self.h, i).decode() for i in range(ns.value)]

    def next(self, cap, seq_stride=80):
        """(STREAM_RECORD [n], packed bases [n, seq_stride]); n = 0 at the end"""
        recs = np.zeros(cap, STREAM_RECORD)
        seq = np.zeros((cap, seq_stride), np.uint8)
        n = C.c_uint32()
        check(lib().gtx_reads_next(self.h, _p(recs), _p(seq), seq_stride, cap, C.byref(n)))
        return recs[:n.value], seq[:n.value]

    def close(self):
        if getattr(self, "h", None):
            lib().gtx_reads_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VcfHeaderRequest(C.Structure):
    _fields_ = [("file_date", C.c_char_p), ("version", C.c_char_p), ("dirty", C.c_int32), ("git_branch", C.c_char_p), ("git_sha1", C.c_char_p),
                ("contig_names", C.POINTER(C.c_char_p)), ("contig_lengths", C.POINTER(C.c_uint32)), ("n_contigs", C.c_uint32),
                ("sample_names", C.POINTER(C.c_char_p)), ("n_samples", C.c_uint32), ("drop_genotypes", C.c_int32)]


def vcf_header(file_date, version, contigs, sample_names, dirty=False, git_branch="", git_sha1="", drop_genotypes=False):
    """gtx_vcf_header: contigs = [(name, length)] -> bytes"""
    names = (C.c_char_p * max(1, len(contigs)))(*[c[0].encode() for c in contigs])
    lens = (C.c_uint32 * max(1, len(contigs)))(*[c[1] for c in contigs])
    samples = (C.c_char_p * max(1, len(sample_names)))(*[s_.encode() for s_ in sample_names])
    rq = VcfHeaderRequest(file_date.encode(), version.encode(), int(dirty), git_branch.encode(), git_sha1.encode(), names, lens, len(contigs),
                          samples, len(sample_names), int(drop_genotypes))
    n = C.c_uint64()
    check(lib().gtx_vcf_header(C.byref(rq), None, 0, C.byref(n)))
    buf = C.create_string_buffer(int(n.value) + 1)
    check(lib().gtx_vcf_header(C.byref(rq), buf, n.value, C.byref(n)))
    return buf.raw[:int(n.value)]


def bgzf_compress(data, level=-1, with_eof=True):
    """gtx_bgzf_compress -> bytes"""
    n = C.c_uint64()
    src = C.create_string_buffer(data, len(data))
    check(lib().gtx_bgzf_compress(src, len(data), level, int(with_eof), None, 0, C.byref(n)))
    out = C.create_string_buffer(int(n.value) + 1)
    check(lib().gtx_bgzf_compress(src, len(data), level, int(with_eof), out, n.value, C.byref(n)))
    return out.raw[:int(n.value)]


def inflate_raw(data, out_len):
    """gtx_inflate_raw -> bytes (raises GtxError on a stream that is not valid or not of that size)"""
    out = C.create_string_buffer(max(out_len, 1))
    check(lib().gtx_inflate_raw(data, len(data), out, out_len))
    return out.raw[:out_len]


def pipeline_run(ctx, paths, n_threads, buf, rec_words, record_slots_per_thread, chunk=65536, region=None):
    """gtx_pipeline_run over BAM files into the accumulator block `buf` (ScoreBuffers) -> its statistics as a dict"""
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    st = PipelineStats()
    check(lib().gtx_pipeline_run(ctx.h, arr, len(paths), n_threads, region.encode() if region else None, chunk, rec_words, record_slots_per_thread,
                                 C.byref(buf), C.byref(st)))
    return {k: getattr(st, k) for k, _ in PipelineStats._fields_}


class RegionJobs:
    """the job array of gtx_regions_run.  regions: dicts with reference (str), region_begin, records [(pos0, ref, [alts], info)],
    d_planes / d_meta / d_items (device pointers as int), plane_stride, n_reads, n_items and optionally region_end, add_all_variants,
    vcf_begin, vcf_end, filter_zero_qual.  Built once (the conversion of the records is the caller's staging), run any number of times."""

    def __init__(self, regions):
        self.n = len(regions)
        self.jobs = (RegionJob * max(self.n, 1))()
        self.keep = []
        for k, r in enumerate(regions):
            recs, keep = records_array(r["records"])
            refb = r["reference"].encode()
            self.keep.extend([recs, keep, refb])
            j = self.jobs[k]
            j.reference, j.reference_len = refb, len(refb)
            j.region_begin, j.region_end = r["region_begin"], r.get("region_end", 0xFFFFFFFF)
            j.records, j.n_records = recs, len(r["records"])
            j.add_all_variants = int(r.get("add_all_variants", False))
            j.d_planes, j.plane_stride, j.d_meta, j.n_reads = r["d_planes"], r["plane_stride"], r["d_meta"], r["n_reads"]
            j.d_items, j.n_items = r["d_items"], r["n_items"]
            j.vcf_begin, j.vcf_end = r.get("vcf_begin", 0), r.get("vcf_end", 0xFFFFFFFF)
            j.filter_zero_qual = int(r.get("filter_zero_qual", False))

    def run(self, sample_names, contig="chr1", device=0, rec_words=8, conn_cap=1 << 16, builders=0, device_threads=0, text_threads=0, params=None):
        """gtx_regions_run -> ([text of job k as bytes, or None where the job failed], statistics as a dict); raises on the first failing job's status"""
        L = lib()
        names = (C.c_char_p * max(1, len(sample_names)))(*[n.encode() for n in sample_names])
        params = params or Params(75, 0, 0, 0, 0, 3840, 0, 0, 0)
        st = RegionsStats()
        rc = L.gtx_regions_run(self.jobs, self.n, C.byref(params), device, contig.encode(), names, len(sample_names), rec_words, conn_cap, builders,
                               device_threads, text_threads, C.byref(st))
        texts = [C.string_at(self.jobs[k].text, self.jobs[k].text_len) if self.jobs[k].text else None for k in range(self.n)]
        self.status = [int(self.jobs[k].status) for k in range(self.n)]
        self.texts = texts  # (kept for a caller that catches the first failing job's error: the other jobs have theirs)
        L.gtx_regions_free(self.jobs, self.n)
        check(rc)
        return texts, {k: getattr(st, k) for k, _ in RegionsStats._fields_ if k != "reserved"}


def tabix_build(vcf_gz, min_shift=0, index_path=None):
    """gtx_tabix_build: <vcf>.tbi (min_shift 0) or a .csi"""
    check(lib().gtx_tabix_build(vcf_gz.encode(), min_shift, index_path.encode() if index_path else None))


def tabix_start(vcf_gz, chrom, begin, end):
    """gtx_tabix_start -> virtual offset, or None when the index knows of no record there"""
    v, any_ = C.c_uint64(), C.c_int()
    check(lib().gtx_tabix_start(vcf_gz.encode(), chrom.encode(), begin, end, C.byref(v), C.byref(any_)))
    return int(v.value) if any_.value else None


def shrink_params(**kw):
    """gtx_shrink_params with the reference's defaults, fields overridden by keyword"""
    p = ShrinkParams()
    lib().gtx_shrink_params_default(C.byref(p))
    for k, v in kw.items():
        assert hasattr(p, k), k
        setattr(p, k, v)
    return p


def bam_shrink(bam_in, intervals, bam_out, params=None):
    """gtx_bam_shrink; intervals: [(chrom, begin, end)] 0-based, both inside.  Returns the stats as a dict."""
    n = len(intervals)
    chroms = (C.c_char_p * n)(*[c.encode() for c, _, _ in intervals])
    begins = (C.c_int32 * n)(*[b for _, b, _ in intervals])
    ends = (C.c_int32 * n)(*[e for _, _, e in intervals])
    st = ShrinkStats()
    check(lib().gtx_bam_shrink(bam_in.encode(), chroms, begins, ends, n, C.byref(params) if params is not None else None, bam_out.encode(), C.byref(st)))
    return {k: int(getattr(st, k)) for k, _ in ShrinkStats._fields_}


def bam_shrink_multi(bam_in, interval_file, bam_out, params=None):
    """gtx_bam_shrink_multi: the intervals come from a file of 'contig first last' lines (1-based)"""
    st = ShrinkStats()
    check(lib().gtx_bam_shrink_multi(bam_in.encode(), interval_file.encode(), C.byref(params) if params is not None else None, bam_out.encode(), C.byref(st)))
    return {k: int(getattr(st, k)) for k, _ in ShrinkStats._fields_}


class VcfRequest(C.Structure):
    _fields_ = [("contig", C.c_char_p), ("sample_names", C.POINTER(C.c_char_p)), ("n_samples", C.c_uint32),
                ("region_begin", C.c_uint32), ("region_end", C.c_uint32), ("filter_zero_qual", C.c_int32),
                ("variant_suffix_id", C.c_char_p), ("gt_cov", C.c_void_p), ("stat_u64", C.c_void_p), ("stat_u32", C.c_void_p),
                ("phred", C.c_void_p), ("calls", C.c_void_p), ("sv_table", C.c_char_p), ("ref_depth", C.c_void_p), ("ref_depth_len", C.c_uint32)]


class Context:
    """gtx_ctx: flat graph + index (host) and, for device >= 0, their copies in HBM"""

    def __init__(self, graph, device=0, max_index_labels=75, is_sv_graph=False, hq_reads=False, force_both=False,
                 is_segment_calling=False, sam_flag_filter=3840, no_second_pass=False, big_record_words=0, exact_pass_mb=0):
        L = lib()
        self.g = {k: np.ascontiguousarray(v) for k, v in graph.items()}
        g = self.g
        has_ev = len(g.get("event_val", ())) > 0
        self.view = GraphView(len(g["ref_order"]), len(g["var_order"]), _p(g["ref_order"]), _p(g["ref_len"]),
                              _p(g["ref_dna_off"]), _p(g["ref_nvar"]), _p(g["ref_first_var"]), _p(g["var_order"]),
                              _p(g["var_len"]), _p(g["var_dna_off"]), _p(g["var_out_ref"]), _p(g["dna"]), len(g["dna"]),
                              _p(g["event_off"]) if has_ev else None, _p(g["event_val"]) if has_ev else None)
        self.params = Params(max_index_labels, int(is_sv_graph), int(hq_reads), int(force_both), int(is_segment_calling),
                             sam_flag_filter, int(no_second_pass), int(big_record_words), int(exact_pass_mb))
        h = C.c_void_p()
        check(L.gtx_ctx_create(C.byref(self.view), C.byref(self.params), device, C.byref(h)))
        self.h = h
        self.device = device
        lay = ScoreLayout()
        check(L.gtx_ctx_score_layout(self.h, C.byref(lay)))
        self.n_hap, self.total_tri, self.total_allele = int(lay.n_hap), int(lay.total_tri), int(lay.total_allele)
        self.total_near = int(lay.total_near)
        self.ref_depth_len = int(lay.ref_depth_len)
        self.hap_order = np.zeros(self.n_hap, np.uint32)
        self.hap_cnum = np.zeros(self.n_hap, np.uint32)
        self.tri_off = np.zeros(self.n_hap, np.uint64)
        self.allele_off = np.zeros(self.n_hap, np.uint64)
        check(L.gtx_ctx_haplotypes(self.h, _p(self.hap_order), _p(self.hap_cnum), _p(self.tri_off), _p(self.allele_off)))
        self.near_last = np.zeros(self.n_hap, np.uint32)  # layout of the near-pair connection counters (d_conn_near)
        self.near_off = np.zeros(self.n_hap, np.uint64)
        check(L.gtx_ctx_near_pairs(self.h, _p(self.near_last), _p(self.near_off)))

    def close(self):
        if getattr(self, "h", None):
            lib().gtx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    def special_positions(self):
        L = lib()
        n = C.c_uint32()
        check(L.gtx_ctx_special_positions(self.h, C.byref(n), None, None, 0))
        rr, ap = np.zeros(n.value, np.uint32), np.zeros(n.value, np.uint32)
        check(L.gtx_ctx_special_positions(self.h, C.byref(n), _p(rr), _p(ap), n.value))
        return rr, ap

    def index_stats(self):
        nk, nl = C.c_uint64(), C.c_uint64()
        check(lib().gtx_index_stats(self.h, C.byref(nk), C.byref(nl)))
        return int(nk.value), int(nl.value)

    def index_get(self, key):
        out = np.zeros(4096, LABEL)
        n = C.c_uint32()
        check(lib().gtx_index_get(self.h, C.c_uint64(key), _p(out), 4096, C.byref(n)))
        return [tuple(int(x) for x in out[i]) for i in range(n.value)]

    def index_dump(self):
        nk, nl = self.index_stats()
        keys, counts, labels = np.zeros(nk, np.uint64), np.zeros(nk, np.uint32), np.zeros(nl, LABEL)
        check(lib().gtx_index_dump(self.h, _p(keys), _p(counts), _p(labels)))
        lab = np.stack([labels["start_index"], labels["end_index"], labels["variant_id"]], axis=1) if nl else np.zeros((0, 3), np.uint32)
        return keys, counts, lab

    def hint_table(self, which):
        """one table of the position-hinted pass as uint32 words (gtx_ctx_hint_table): 0 position flags, 1 reference planes,
        2 tail sites, 3 / 4 half-key filters, 5 allele windows (8 words each), 6 first window | count << 24 per reference node"""
        n = C.c_uint64()
        check(lib().gtx_ctx_hint_table(self.h, which, None, 0, C.byref(n)))
        out = np.zeros(n.value // 4, np.uint32)
        if n.value:
            check(lib().gtx_ctx_hint_table(self.h, which, _p(out), n.value, C.byref(n)))
        return out

    def profile(self):
        out = np.zeros(32, np.uint64)
        check(lib().gtx_ctx_profile(self.h, _p(out)))
        return out

    def profile_log(self, cap=1 << 16):
        """the general pass' task log of the profiling build: rows of 16 words (include/gtx.h: gtx_ctx_profile_log)"""
        out = np.zeros((cap, 16), np.uint64)
        n = C.c_uint64()
        check(lib().gtx_ctx_profile_log(self.h, _p(out), cap, C.byref(n)))
        return out[:n.value]

    def big_records(self):
        """(used part of the big-record arena as a host array, number of tasks the last align batch sent through the
        second pass)"""
        ptr, cap, used, tasks = C.c_void_p(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib().gtx_ctx_big_records(self.h, C.byref(ptr), C.byref(cap), C.byref(used), C.byref(tasks)))
        return download(ptr.value or 0, np.uint32, int(used.value)), int(tasks.value)

    def exact_pass_tasks(self):
        """(tasks of the last align batch that went through the exact pass with a small part of its slab, again with a large part,
        again with the whole slab, tasks that keep a table-overflow status even so)"""
        out = (C.c_uint64 * 4)()
        check(lib().gtx_ctx_exact_pass_tasks(self.h, out))
        return int(out[0]), int(out[1]), int(out[2]), int(out[3])

    def pass_times(self):
        """(ms of the express / general / HBM-table pass of the last align batch, tasks handed to the general pass);
        the first call only arms the timing"""
        ms = (C.c_float * 3)()
        q = C.c_uint32()
        check(lib().gtx_ctx_pass_times(self.h, ms, C.byref(q)))
        return [float(x) for x in ms], int(q.value)

    def kernel_times(self):
        """[(kernel, ms, forward tasks completed)] of the four launches of the last timed align batch"""
        ms, tasks = (C.c_float * 4)(), (C.c_uint32 * 4)()
        check(lib().gtx_ctx_kernel_times(self.h, ms, tasks))
        names = ("gtx_align_hinted_kernel", "gtx_align_express4_kernel", "gtx_align_kernel", "gtx_align_big_kernel")
        return [(n, float(m), int(t)) for n, m, t in zip(names, ms, tasks)]

    def rewind_big_records(self):
        check(lib().gtx_ctx_big_records_rewind(self.h, None))

    def vcf_records(self, contig, sample_names, gt_cov, stat_u64, stat_u32, phred, calls, region_begin=0, region_end=0xFFFFFFFF,
                    filter_zero_qual=False, variant_suffix_id=None, sv_table=None, ref_depth=None):
        """gtx_vcf_records: the VCF records (column line first) of the region's variant sites as bytes; all arrays are host copies.
        SV graphs: sv_table = the text of gtx_graph_sv_table, ref_depth = the FINALISED reference-depth rows [n_samples, ref_depth_len + 1]"""
        names = (C.c_char_p * max(1, len(sample_names)))(*[n.encode() for n in sample_names])
        keep = [np.ascontiguousarray(gt_cov, np.uint32), np.ascontiguousarray(stat_u64, np.uint64), np.ascontiguousarray(stat_u32, np.uint32),
                np.ascontiguousarray(phred, np.uint8), np.ascontiguousarray(calls, SAMPLE_CALL)]
        if ref_depth is not None:
            keep.append(np.ascontiguousarray(ref_depth, np.uint32))
            assert keep[-1].size == len(sample_names) * (self.ref_depth_len + 1)
        rq = VcfRequest(contig.encode(), names, len(sample_names), region_begin, region_end, int(filter_zero_qual),
                        variant_suffix_id.encode() if variant_suffix_id else None, *[C.c_void_p(a.ctypes.data) for a in keep[:5]],
                        sv_table.encode() if sv_table is not None else None, C.c_void_p(keep[5].ctypes.data) if ref_depth is not None else None,
                        self.ref_depth_len if ref_depth is not None else 0)
        n = C.c_uint64()
        check(lib().gtx_vcf_records(self.h, C.byref(rq), None, C.c_uint64(0), C.byref(n)))
        buf = C.create_string_buffer(int(n.value) + 1)
        check(lib().gtx_vcf_records(self.h, C.byref(rq), buf, C.c_uint64(n.value), C.byref(n)))
        return buf.raw[:int(n.value)]

    def vcf_records_final(self, contig, sample_names, gt_cov, stat_u64, stat_u32, phred, calls, region_begin=0, region_end=0xFFFFFFFF,
                          filter_zero_qual=False, variant_suffix_id=None, no_variant_overlapping=False, no_filter_bad_alts=False):
        """gtx_vcf_records_final: the records (column line first) of the file genotype() ends with -- vcf_merge_and_break with the
        variants broken down"""
        names = (C.c_char_p * max(1, len(sample_names)))(*[n.encode() for n in sample_names])
        keep = [np.ascontiguousarray(gt_cov, np.uint32), np.ascontiguousarray(stat_u64, np.uint64), np.ascontiguousarray(stat_u32, np.uint32),
                np.ascontiguousarray(phred, np.uint8), np.ascontiguousarray(calls, SAMPLE_CALL)]
        rq = VcfRequest(contig.encode(), names, len(sample_names), region_begin, region_end, int(filter_zero_qual),
                        variant_suffix_id.encode() if variant_suffix_id else None, *[C.c_void_p(a.ctypes.data) for a in keep], None, None, 0)
        n = C.c_uint64()
        args = (self.h, C.byref(rq), int(no_variant_overlapping), int(no_filter_bad_alts))
        check(lib().gtx_vcf_records_final(*args, None, C.c_uint64(0), C.byref(n)))
        buf = C.create_string_buffer(int(n.value) + 1)
        check(lib().gtx_vcf_records_final(*args, buf, C.c_uint64(n.value), C.byref(n)))
        return buf.raw[:int(n.value)]

    def vcf_sites(self, contig, n_samples, gt_cov, stat_u64, stat_u32, phred, calls, ph_rows):
        """gtx_vcf_sites: the sites-only VCF records (column line first) of the alleles the next iteration keeps, with their GT_ID /
        GT_ANTI_HAPLOTYPE / GT_HAPLOTYPE tags; ph_rows = what phase_flags returned"""
        names = (C.c_char_p * max(1, n_samples))(*[b"S%d" % i for i in range(n_samples)])
        keep = [np.ascontiguousarray(gt_cov, np.uint32), np.ascontiguousarray(stat_u64, np.uint64), np.ascontiguousarray(stat_u32, np.uint32),
                np.ascontiguousarray(phred, np.uint8), np.ascontiguousarray(calls, SAMPLE_CALL)]
        rq = VcfRequest(contig.encode(), names, n_samples, 0, 0xFFFFFFFF, 0, None, *[C.c_void_p(a.ctypes.data) for a in keep], None, None, 0)
        ph = np.zeros(len(ph_rows), PHASE_ENTRY)
        for k, f in enumerate(("hap1", "allele1", "hap2", "allele2", "flags")):
            ph[f] = np.asarray(ph_rows)[:, k] if len(ph_rows) else 0
        n = C.c_uint64()
        check(lib().gtx_vcf_sites(self.h, C.byref(rq), _p(ph), C.c_uint64(len(ph)), None, C.c_uint64(0), C.byref(n)))
        buf = C.create_string_buffer(int(n.value) + 1)
        check(lib().gtx_vcf_sites(self.h, C.byref(rq), _p(ph), C.c_uint64(len(ph)), buf, C.c_uint64(n.value), C.byref(n)))
        return buf.raw[:int(n.value)]

    def phase_flags(self, n_samples, gt_cov, conn_log, n_conn, conn_near=None):
        """gtx_phase_flags: rows (hap1, allele1, hap2, allele2, flags) as an int array [n, 5]"""
        gt_cov = np.ascontiguousarray(gt_cov, np.uint32)
        conn_log = np.ascontiguousarray(conn_log, np.uint32)
        if conn_near is not None:
            conn_near = np.ascontiguousarray(conn_near, np.uint32)
            assert len(conn_near) == n_samples * self.total_near
        n = C.c_uint64()
        cap = 1024
        while True:
            out = np.zeros(cap, PHASE_ENTRY)
            rc = lib().gtx_phase_flags(self.h, n_samples, _p(gt_cov), _p(conn_log), C.c_uint64(n_conn),
                                       _p(conn_near) if conn_near is not None else None, _p(out), C.c_uint64(cap), C.byref(n))
            if rc == 5 and n.value > cap:  # GTX_ERR_CAPACITY
                cap = int(n.value)
                continue
            check(rc)
            out = out[:n.value]
            return np.stack([out["hap1"], out["allele1"], out["hap2"], out["allele2"], out["flags"]], axis=1).astype(np.int64)

    def error_count(self):
        n = C.c_uint32()
        check(lib().gtx_ctx_error_count(self.h, C.byref(n)))
        return int(n.value)


class Stream:
    """gtx_stream: which records align, which reuse the previous alignment, which pairs / reads get scored"""

    def __init__(self, params, n_read_groups=1):
        h = C.c_void_p()
        check(lib().gtx_stream_create(C.byref(params), n_read_groups, C.byref(h)))
        self.h = h

    def __del__(self):
        if getattr(self, "h", None):
            lib().gtx_stream_destroy(self.h)
            self.h = None

    def set_planes(self, plane_stride):
        """gtx_stream_set_planes: push() then returns plane rows of that pitch (0: BAM nibble rows again)"""
        check(lib().gtx_stream_set_planes(self.h, plane_stride))
        self.plane_stride = plane_stride

    def push(self, recs, seq):
        """recs: STREAM_RECORD array, seq: [n, stride] packed bases -> (align_seq, align_meta, items)"""
        recs = np.ascontiguousarray(recs, STREAM_RECORD)
        seq = np.ascontiguousarray(seq, np.uint8)
        n, stride = seq.shape
        a_seq = np.zeros((n, getattr(self, "plane_stride", 0) or stride), np.uint8)
        a_meta = np.zeros(n, READ_META)
        items = np.zeros(n, SCORE_ITEM)
        na, ni = C.c_uint32(), C.c_uint32()
        check(lib().gtx_stream_push(self.h, _p(recs), _p(seq), stride, n, _p(a_seq), _p(a_meta), n, C.byref(na), _p(items), n,
                                    C.byref(ni)))
        return a_seq[:na.value], a_meta[:na.value], items[:ni.value]

    def set_coverage(self, avg_cov_by_readlen):
        a = np.ascontiguousarray(avg_cov_by_readlen, np.float64)
        check(lib().gtx_stream_set_coverage(self.h, _p(a), len(a)))

    def finish(self):
        """end of the stream -> leftover items (SV calling), parked reads are forgotten"""
        cap = max(self.counts()["parked"], 1)
        items = np.zeros(cap, SCORE_ITEM)
        ni = C.c_uint32()
        check(lib().gtx_stream_finish(self.h, _p(items), cap, C.byref(ni)))
        return items[:ni.value]

    def counts(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib().gtx_stream_counts(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(records=int(a.value), duplicated=int(b.value), parked=int(c.value))


REC_WIDE, WIDE_MASK_WORDS = 0x40000000, 80  # include/gtx.h: GTX_REC_WIDE, GTX_WIDE_MASK_WORDS


TASK_HAS_VARIANTS, TASK_COMPACT, COMPACT_WORDS = 1, 2, 8
WORK_HEADER_WORDS = 4  # GTX_WORK_HEADER_WORDS: gtx_align_batch_planes_triaged's queue starts behind them
TRIAGE_ITEMS_ARE_READS = 1  # GTX_TRIAGE_ITEMS_ARE_READS
REPLAY_ENTRY = np.dtype([("item", np.uint32), ("cell", np.uint32), ("order_eps", np.uint32), ("mask_lo", np.uint32), ("mask_hi", np.uint32), ("pad", np.uint32)])


def merge_compact(records, compact, task_flags, n_reads, rec_words):
    """the records of gtx_align_batch_planes_compact as gtx_align_batch would have left them: a task whose byte of the side array
    carries GTX_TASK_COMPACT has its record in `compact` (8 words per read) and an untouched slot -> a copy with those records in
    their slots (host arrays; what parse_records and the tests' comparisons read)"""
    out = np.array(records, np.uint32).reshape(n_reads * 2, rec_words)
    comp = np.asarray(compact, np.uint32).reshape(-1, COMPACT_WORDS)[:n_reads]
    fl = np.asarray(task_flags, np.uint8).reshape(-1)[:2 * n_reads]
    which = np.nonzero(fl[0::2] & TASK_COMPACT)[0]
    assert not (fl[1::2] & TASK_COMPACT).any(), "only forward records are compact"
    out[2 * which, :COMPACT_WORDS] = comp[which]
    out[2 * which, COMPACT_WORDS:] = 0
    return out.reshape(-1)


def parse_records(words, n_reads, rec_words, hap_order, big_records=None):
    """decode gtx_align_batch records into the same structure tests/oracle_lib.parse_path_stream returns
    (plus 'status', with the GTX_ST_EXTERNAL bit removed); Path::var_order is looked up through hap_order.
    big_records: the context's big-record arena (needed when a record carries GTX_ST_EXTERNAL)"""
    words = np.asarray(words, np.uint32).reshape(n_reads * 2, rec_words)
    out = []
    for i in range(n_reads):
        pair = []
        for o in range(2):
            w = words[2 * i + o]
            npaths, status = int(w[0]) & 0xFFFF, int(w[0]) >> 16
            longest, rlen = int(w[1]) & 0xFFFF, (int(w[1]) >> 16) & 0x3FFF  # (bit 31: REC_HAS_VARIANTS, bit 30: REC_WIDE)
            mw = WIDE_MASK_WORDS if int(w[1]) & REC_WIDE else 2
            k = 2
            if status & ST_EXTERNAL:
                w, k = big_records, int(w[2])
                status &= ~ST_EXTERNAL
            paths = []
            for _ in range(npaths):
                st, en, rsre, mmnv = int(w[k]), int(w[k + 1]), int(w[k + 2]), int(w[k + 3])
                k += 4
                vs = []
                for _v in range(mmnv >> 16):
                    hap = int(w[k])
                    alleles = tuple(32 * x + a for x in range(mw) if w[k + 1 + x] for a in range(32) if (int(w[k + 1 + x]) >> a) & 1)
                    k += 1 + mw
                    vs.append((int(hap_order[hap]), alleles))
                paths.append(dict(start=st, end=en, rs=rsre & 0xFFFF, re=rsre >> 16, mm=mmnv & 0xFFFF, vars=vs))
            pair.append(dict(longest=longest, paths=paths, status=status, read_len=rlen))
        out.append(tuple(pair))
    return out
