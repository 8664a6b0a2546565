"""ctypes binding of libkcount_mi355.so (include/kcount_mi355.h).  No fallback of any kind."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KC_OK = 0
KC_ERR_CAPACITY = -6
KC_ERR_BAD_BASE = -7
KC_FLAG_TIME_KERNELS = 1
KC_FLAG_REFERENCE_OWNER = 2
KC_FLAG_SHARD_BUCKETS = 4
KC_FLAG_WIRE_UNITS = 8


class KcError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        L = lib()
        msg = L.kc_error_string(status).decode()
        detail = L.kc_last_error().decode()
        super().__init__("%s: %s (%d)%s" % (where, msg, status, (" -- " + detail) if detail and status in (-3, -4, -5, -6) else ""))


class kc_config(C.Structure):
    _fields_ = [("kmer_len", C.c_int32), ("qual_offset", C.c_int32), ("dmin_thres", C.c_int32), ("device", C.c_int32),
                ("rank_me", C.c_int32), ("rank_n", C.c_int32), ("max_elems", C.c_uint64), ("flags", C.c_uint32),
                ("reserved", C.c_uint32), ("max_kmers_buffered", C.c_uint64)]


class kc_stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("num_reads", "num_bases", "raw_kmers", "kmers_inserted", "num_unique", "num_purged",
                                           "total_kmers", "sum_counts", "num_dropped", "capacity", "num_gpu_calls",
                                           "table_bytes")]


class kc_result(C.Structure):
    _fields_ = [("n", C.c_uint64), ("num_longs", C.c_int32), ("reserved", C.c_int32), ("d_keys", C.c_void_p),
                ("d_counts", C.c_void_p), ("d_left", C.c_void_p), ("d_right", C.c_void_p)]


class kc_tuning(C.Structure):
    _fields_ = [("mode", C.c_uint32), ("writers", C.c_uint32), ("p1", C.c_uint32), ("p2", C.c_uint32), ("slots", C.c_uint32),
                ("chunk1", C.c_uint32), ("chunk2", C.c_uint32), ("chain1_max", C.c_uint32), ("chain2_max", C.c_uint32),
                ("arena1", C.c_uint32), ("ovf_capacity", C.c_uint64)]


class kc_kernel_time(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


class kc_synth_params(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("num_genomes", C.c_uint32), ("read_len", C.c_uint32), ("min_genome_len", C.c_uint64),
                ("max_genome_len", C.c_uint64), ("sub_error_rate", C.c_double), ("lowq_rate", C.c_double),
                ("n_rate", C.c_double), ("abundance_sigma", C.c_double)]


# every symbol include/kcount_mi355.h declares: (restype, argtypes)
SYMBOLS = {
    "kc_abi_version": (C.c_int, []),
    "kc_error_string": (C.c_char_p, [C.c_int]),
    "kc_last_error": (C.c_char_p, []),
    "kc_device_count": (C.c_int, []),
    "kc_num_longs": (C.c_int, [C.c_int]),
    "kc_record_longs": (C.c_int, [C.c_int]),
    "kc_owner": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "kc_owner_reference": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "kc_create": (C.c_void_p, [C.POINTER(kc_config), C.POINTER(C.c_int)]),
    "kc_destroy": (None, [C.c_void_p]),
    "kc_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kc_reset": (C.c_int, [C.c_void_p, C.c_int]),
    "kc_submit_reads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]),
    "kc_submit_packed_reads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]),
    "kc_fastq_to_packed": (C.c_int, [C.c_char_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_uint64)]),
    "kc_submit_seq_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]),
    "kc_extract_partition": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p,
                                       C.c_uint64, C.c_void_p]),
    "kc_extract_partition_seq_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]),
    "kc_insert_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "kc_shard_extract": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]),
    "kc_shard_extract_seq_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]),
    "kc_shard_reserve": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "kc_shard_commit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "kc_shard_owner": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "kc_wire_unit": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "kc_partition_owner": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "kc_insert_record_pieces": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]),
    "kc_shard_capacity": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "kc_build_supermers": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint32), C.c_void_p]),
    "kc_submit_packed_supermers": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]),
    "kc_flush": (C.c_int, [C.c_void_p]),
    "kc_finalize": (C.c_int, [C.c_void_p, C.POINTER(kc_result)]),
    "kc_copy_results": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kc_begin_ctg_kmers": (C.c_int, [C.c_void_p, C.c_uint64]),
    "kc_submit_ctg_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]),
    "kc_ctg_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "kc_arena_probe_rate": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "kc_copy_results_entries": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "kc_lookup": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kc_dump_table": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]),
    "kc_get_stats": (C.c_int, [C.c_void_p, C.POINTER(kc_stats)]),
    "kc_set_tuning": (C.c_int, [C.c_void_p, C.POINTER(kc_tuning)]),
    "kc_get_kernel_times": (C.c_int, [C.c_void_p, C.POINTER(kc_kernel_time), C.c_int, C.POINTER(C.c_int)]),
    "kc_clear_kernel_times": (C.c_int, [C.c_void_p]),
    "kc_synth_default_params": (None, [C.POINTER(kc_synth_params)]),
    "kc_synth_reads_host": (C.c_int, [C.POINTER(kc_synth_params), C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kc_synth_reads_device": (C.c_int, [C.c_void_p, C.POINTER(kc_synth_params), C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
}


def lib_path():
    # KC_LIB: an alternative build of the same library (tuning experiments only)
    return os.environ.get("KC_LIB") or os.path.join(_HERE, "csrc", "libkcount_mi355.so")


def lib():
    """Load the HIP library.  Raises if it has not been built: there is no other implementation."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or make -C mhm2_kmer_analysis_v2_amd/csrc); there is no CPU fallback" % p)
        L = C.CDLL(p)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


def check(status, where):
    if status != KC_OK:
        raise KcError(status, where)
