"""ctypes declarations for libcrass_hip.so — one entry per function in include/crass_hip.h.

The library is the product: if it cannot be loaded this module raises, it never falls back
to a CPU implementation."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcrass_hip.so")

u8p, u16p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
charp = C.POINTER(C.c_char)


class Params(C.Structure):
    _fields_ = [("lowDRsize", C.c_uint32), ("highDRsize", C.c_uint32), ("lowSpacerSize", C.c_uint32),
                ("highSpacerSize", C.c_uint32), ("searchWindowLength", C.c_uint32),
                ("minNumRepeats", C.c_uint32), ("kmer_clust_size", C.c_int32)]


class Reads(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("packed", C.c_void_p), ("stride_words", C.c_uint32),
                ("word_off", C.c_void_p), ("uniform_len", C.c_uint32), ("lengths", C.c_void_p),
                ("n_exceptions", C.c_uint64), ("exc_read", C.c_void_p), ("exc_off", C.c_void_p),
                ("exc_bytes", C.c_void_p), ("header_id", C.c_void_p), ("read_index_base", C.c_uint64)]


class Candidates(C.Structure):
    _fields_ = [("n", C.c_uint64), ("read_idx", u64p), ("low_lexi", u8p), ("repeat_len", u32p), ("n_ss", u32p),
                ("ss_off", u64p), ("ss_pool", u32p), ("dr_stride", C.c_uint32), ("dr_len", u16p),
                ("dr_chars", charp), ("max_read_len", C.c_uint32)]


class MergeView(C.Structure):
    _fields_ = [("n_tokens", C.c_uint32), ("tok_chars", charp), ("tok_off", u64p), ("n_candidates", C.c_uint64),
                ("cand_token", u32p), ("n_groups", C.c_uint32), ("grp_tokens", u32p), ("grp_off", u64p),
                ("n_patterns", C.c_uint32), ("pat_chars", charp), ("pat_off", u64p), ("pat_group", u32p),
                ("next_free_gid", C.c_int32)]


class Distinct(C.Structure):
    _fields_ = [("n_distinct", C.c_uint64), ("dr_stride", C.c_uint32), ("dr_len", u16p), ("dr_chars", charp),
                ("n_candidates", C.c_uint64), ("cand_distinct", u32p)]


class Recruits(C.Structure):
    _fields_ = [("n", C.c_uint64), ("read_idx", u64p), ("low_lexi", u8p), ("start", u32p), ("end", u32p),
                ("dr_stride", C.c_uint32), ("dr_len", u16p), ("dr_chars", charp), ("token", u32p)]


class Counters(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_exceptions", C.c_uint64), ("n_filter_survivors", C.c_uint64),
                ("n_pass1_found", C.c_uint64), ("n_pass2_found", C.c_uint64), ("n_patterns", C.c_uint32),
                ("ac_states", C.c_uint32), ("used_fast_filter", C.c_uint32), ("used_lds_automaton", C.c_uint32),
                ("ms_filter", C.c_float), ("ms_compact", C.c_float), ("ms_survivor", C.c_float),
                ("ms_pass1_total", C.c_float), ("ms_recruit", C.c_float), ("ms_recruit_finish", C.c_float),
                ("ms_pass2_total", C.c_float), ("ms_merge_host", C.c_float), ("ms_sink_host", C.c_float),
                ("bytes_reads_device", C.c_uint64), ("anchor_keys", C.c_uint32), ("anchor_table_kind", C.c_uint32),
                ("used_device_merge", C.c_uint32), ("ms_merge_device", C.c_float),
                ("n_merge_fallbacks", C.c_uint32), ("last_fallback_bits", C.c_uint32), ("n_bound_overflows", C.c_uint32 * 4),
                ("used_device_view", C.c_uint32), ("n_view_fallbacks", C.c_uint32)]

    def asdict(self):
        d = {f[0]: getattr(self, f[0]) for f in self._fields_}
        d["n_bound_overflows"] = list(self.n_bound_overflows)
        return d


class DistinctDev(C.Structure):
    _fields_ = [("n_distinct", C.c_uint64), ("dr_stride", C.c_uint32), ("d_chars", C.c_void_p), ("d_len", C.c_void_p)]


ERR_OVERFLOW = 8          # CRASS_ERR_OVERFLOW (include/crass_hip.h)


class Exchange(C.Structure):
    _fields_ = [("d_send", C.c_void_p), ("send_bytes", C.c_uint64), ("slot_bytes", C.c_uint32), ("cap_rows", C.c_uint64)]


class Packed(C.Structure):
    _fields_ = [("reads", Reads), ("owner", C.c_void_p)]


class Fastx(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("seq", u8p), ("seq_off", u64p), ("name", u8p), ("name_off", u64p),
                ("comment", u8p), ("comment_off", u64p), ("has_comment", u8p), ("qual", u8p), ("qual_off", u64p),
                ("has_qual", u8p), ("header_id", u64p), ("max_len", C.c_uint32), ("last_ret", C.c_int32),
                ("name_index", u64p), ("name_index_cap", C.c_uint64)]


class SynthSpec(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("read_len", C.c_uint32), ("n_dr", C.c_uint32), ("dr_len_min", C.c_uint32),
                ("dr_len_max", C.c_uint32), ("spacer_len_min", C.c_uint32), ("spacer_len_max", C.c_uint32),
                ("crispr_per_million", C.c_uint32), ("gc_classes", C.c_uint32),
                ("array_min_repeats", C.c_uint32), ("array_max_repeats", C.c_uint32)]


class ConsInput(C.Structure):
    _fields_ = [("seqs", C.c_void_p), ("seq_off", C.c_void_p), ("n_reads", C.c_uint64),
                ("n_rec", C.c_uint64), ("rec_read", C.c_void_p), ("rec_lowlexi", C.c_void_p), ("rec_token", C.c_void_p),
                ("rec_nss", C.c_void_p), ("rec_ss_off", C.c_void_p), ("ss_pool", C.c_void_p),
                ("n_tokens", C.c_uint32), ("tok_chars", C.c_void_p), ("tok_off", C.c_void_p),
                ("n_groups", C.c_uint32), ("grp_tokens", C.c_void_p), ("grp_off", C.c_void_p),
                ("max_read_len", C.c_uint32)]


class ConsCounters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("n_groups_parsed", "n_ksw_launches", "n_ksw_alignments", "n_placements", "n_flips",
                                          "n_true_drs", "n_sw_tasks", "n_partials_added")]

    def asdict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class ConsView(C.Structure):
    _fields_ = [("error", C.c_int32), ("next_free_gid", C.c_int32), ("n_tokens", C.c_uint32),
                ("tok_chars", C.POINTER(C.c_char)), ("tok_off", C.POINTER(C.c_uint64)),
                ("n_groups", C.c_uint32), ("grp_gid", C.POINTER(C.c_int32)), ("dr_chars", C.POINTER(C.c_char)),
                ("dr_off", C.POINTER(C.c_uint64)), ("grp_tokens", C.POINTER(C.c_uint32)), ("grp_off", C.POINTER(C.c_uint64)),
                ("n_rec", C.c_uint64), ("rec_alive", C.POINTER(C.c_uint8)), ("rec_rc", C.POINTER(C.c_uint8)),
                ("rec_token", C.POINTER(C.c_uint32)), ("rec_nss", C.POINTER(C.c_uint32)), ("rec_ss_off", C.POINTER(C.c_uint64)),
                ("ss_pool", C.POINTER(C.c_uint32)), ("tokread_off", C.POINTER(C.c_uint64)), ("tokread_idx", C.POINTER(C.c_uint64)),
                ("tok_has_list", C.POINTER(C.c_uint8)), ("counters", ConsCounters)]


class GraphInput(C.Structure):
    _fields_ = [("n_groups", C.c_uint32), ("gid", C.c_void_p), ("dr_chars", C.c_void_p), ("dr_off", C.c_void_p), ("grp_rec_off", C.c_void_p),
                ("n_rec", C.c_uint64), ("hdr_chars", C.c_void_p), ("hdr_off", C.c_void_p), ("com_chars", C.c_void_p), ("com_off", C.c_void_p),
                ("seq_chars", C.c_void_p), ("seq_off", C.c_void_p), ("rec_nss", C.c_void_p), ("rec_ss_off", C.c_void_p), ("ss_pool", C.c_void_p)]


class OutputOpts(C.Structure):
    _fields_ = [("out_dir", C.c_char_p), ("timestamp", C.c_char_p), ("command_line", C.c_char_p), ("cwd", C.c_char_p),
                ("log_to_screen", C.c_int32), ("cov_cutoff", C.c_int32), ("node_kmer", C.c_int32), ("show_singles", C.c_int32),
                ("long_description", C.c_int32)]


class OutputsView(C.Structure):
    _fields_ = [("n_files", C.c_uint32), ("name", C.POINTER(C.c_char_p)), ("data", C.POINTER(C.c_void_p)), ("size", u64p),
                ("n_groups_kept", C.c_uint32), ("kept_gid", C.POINTER(C.c_int32)), ("stdout_text", C.c_char_p)]


# every exported symbol of include/crass_hip.h: name -> (restype, argtypes)
SYMBOLS = {
    "crass_hip_abi_version": (C.c_int, []),
    "crass_default_params": (None, [C.POINTER(Params)]),
    "crass_hip_create": (C.c_int, [C.POINTER(Params), C.c_int, C.POINTER(C.c_void_p)]),
    "crass_hip_set_stage_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "crass_hip_reload_env": (C.c_int, [C.c_void_p]),
    "crass_hip_set_timing_focus": (C.c_int, [C.c_void_p, C.c_uint]),
    "crass_hip_stream_wait_event": (C.c_int, [C.c_void_p, C.c_void_p]),
    "crass_hip_destroy": (None, [C.c_void_p]),
    "crass_hip_strerror": (C.c_char_p, [C.c_int]),
    "crass_hip_last_hip_error": (C.c_int, [C.c_void_p]),
    "crass_hip_load_reads": (C.c_int, [C.c_void_p, C.POINTER(Reads)]),
    "crass_hip_attach_device_reads": (C.c_int, [C.c_void_p, C.POINTER(Reads)]),
    "crass_hip_seed_scan": (C.c_int, [C.c_void_p]),
    "crass_hip_get_candidates": (C.c_int, [C.c_void_p, C.POINTER(Candidates)]),
    "crass_hip_merge": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64]),
    "crass_hip_get_distinct": (C.c_int, [C.c_void_p, C.POINTER(Distinct)]),
    "crass_hip_merge_distinct": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64]),
    "crass_hip_exchange_setup": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(Exchange)]),
    "crass_hip_merge_gathered": (C.c_int, [C.c_void_p, C.c_void_p]),
    "crass_hip_exchange_needed_rows": (C.c_uint64, [C.c_void_p]),
    "crass_hip_exchange_set_deferred": (C.c_int, [C.c_void_p, C.c_int]),
    "crass_hip_exchange_rows_for": (C.c_uint64, [C.c_uint64]),
    "crass_hip_get_distinct_device": (C.c_int, [C.c_void_p, C.POINTER(DistinctDev)]),
    "crass_hip_merge_distinct_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64]),
    "crass_hip_get_merge": (C.c_int, [C.c_void_p, C.POINTER(MergeView)]),
    "crass_merge_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.POINTER(C.c_void_p)]),
    "crass_merge_rebuild": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                      C.c_uint32, C.POINTER(C.c_void_p)]),
    "crass_merge_get": (C.c_int, [C.c_void_p, C.POINTER(MergeView)]),
    "crass_merge_destroy": (None, [C.c_void_p]),
    "crass_hip_set_patterns": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), u32p, C.c_uint32]),
    "crass_hip_recruit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "crass_hip_get_recruits": (C.c_int, [C.c_void_p, C.POINTER(Recruits)]),
    "crass_hip_levenshtein_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "crass_hip_get_counters": (C.c_int, [C.c_void_p, C.POINTER(Counters)]),
    "crass_hip_stream": (C.c_void_p, [C.c_void_p]),
    "crass_pack_reads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(Packed)]),
    "crass_free_packed": (None, [C.POINTER(Packed)]),
    "crass_read_fastx": (C.c_int, [C.c_char_p, C.POINTER(Fastx)]),
    "crass_free_fastx": (None, [C.POINTER(Fastx)]),
    "crass_hip_consensus": (C.c_int, [C.POINTER(Params), C.c_int, C.POINTER(ConsInput), C.POINTER(C.c_void_p)]),
    "crass_hip_consensus_view": (C.c_int, [C.c_void_p, C.POINTER(ConsView)]),
    "crass_hip_consensus_free": (None, [C.c_void_p]),
    "crass_fastx_find": (C.c_uint64, [C.POINTER(Fastx), C.c_char_p, C.c_uint64]),
    "crass_index_fastx": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "crass_index_fastx_files": (C.c_int, [C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_void_p)]),
    "crass_fastx_index_reads": (C.c_int, [C.c_void_p, C.POINTER(Reads), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "crass_fastx_index_fetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Fastx)]),
    "crass_fastx_index_free": (None, [C.c_void_p]),
    "crass_fastx_index_drop_text": (None, [C.c_void_p]),
    "crass_name_table_create": (C.c_void_p, []),
    "crass_name_table_destroy": (None, [C.c_void_p]),
    "crass_name_table_reserve": (None, [C.c_void_p, C.c_uint64]),
    "crass_name_table_first": (C.c_uint64, [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64]),
    "crass_fastx_stream_open": (C.c_int, [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "crass_fastx_stream_next": (C.c_int, [C.c_void_p, C.POINTER(Fastx)]),
    "crass_fastx_stream_reads_done": (C.c_uint64, [C.c_void_p]),
    "crass_fastx_stream_max_len": (C.c_uint32, [C.c_void_p]),
    "crass_fastx_stream_close": (None, [C.c_void_p]),
    "crass_hip_set_host_view": (C.c_int, [C.c_void_p, C.c_int]),
    "crass_hip_group_create": (C.c_int, [C.POINTER(Params), C.POINTER(C.c_int), C.c_int, C.c_uint, C.POINTER(C.c_void_p)]),
    "crass_hip_group_destroy": (None, [C.c_void_p]),
    "crass_hip_group_size": (C.c_int, [C.c_void_p]),
    "crass_hip_group_rccl_ranks": (C.c_int, [C.c_void_p]),
    "crass_hip_group_last_error": (C.c_char_p, []),
    "crass_hip_group_ctx": (C.c_void_p, [C.c_void_p, C.c_int]),
    "crass_hip_group_load_reads": (C.c_int, [C.c_void_p, C.POINTER(Reads)]),
    "crass_hip_group_seed_scan": (C.c_int, [C.c_void_p]),
    "crass_hip_group_merge": (C.c_int, [C.c_void_p]),
    "crass_hip_group_recruit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "crass_hip_group_set_patterns": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), u32p, C.c_uint32]),
    "crass_hip_group_step": (C.c_int, [C.c_void_p]),
    "crass_hip_group_get_candidates": (C.c_int, [C.c_void_p, C.POINTER(Candidates)]),
    "crass_hip_group_get_merge": (C.c_int, [C.c_void_p, C.POINTER(MergeView)]),
    "crass_hip_group_get_recruits": (C.c_int, [C.c_void_p, C.POINTER(Recruits)]),
    "crass_build_outputs": (C.c_int, [C.POINTER(GraphInput), C.POINTER(OutputOpts), C.POINTER(C.c_void_p)]),
    "crass_outputs_get": (C.c_int, [C.c_void_p, C.POINTER(OutputsView)]),
    "crass_outputs_write": (C.c_int, [C.c_void_p, C.c_char_p]),
    "crass_outputs_free": (None, [C.c_void_p]),
    "crass_synth_default": (None, [C.POINTER(SynthSpec)]),
    "crass_synth_packed": (C.c_int, [C.POINTER(SynthSpec), C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]),
    "crass_unpack_ascii": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p]),
}

ABI_VERSION = 3        # CRASS_HIP_ABI_VERSION of include/crass_hip.h these struct layouts mirror
_lib = None


def load():
    """Load libcrass_hip.so and bind every symbol; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "crass_amd: %s is missing. Build it with `python -m crass_amd.build` (needs hipcc). "
            "There is no CPU fallback for the search path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    if lib.crass_hip_abi_version() != ABI_VERSION:
        raise RuntimeError("crass_amd: %s has ABI version %d, these bindings were written for %d (struct layouts differ): rebuild"
                           % (LIB_PATH, lib.crass_hip_abi_version(), ABI_VERSION))
    _lib = lib
    return lib
