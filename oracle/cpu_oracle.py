"""ctypes binding of oracle/liborc.so (kcount_oracle.c) -- TEST INFRASTRUCTURE ONLY.

The C file restates the reference CPU kcount (src/kcount/kcount_cpu.cpp,
src/kmer.cpp, src/hash_funcs.c; citations in the C source).  This module only
marshals numpy arrays in and out of it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u64p = C.POINTER(C.c_uint64)
u16p = C.POINTER(C.c_uint16)


def build(force=False):
    if os.environ.get("ORC_LIB"):  # e.g. the sanitizer build: make -C oracle liborc_asan.so
        return os.environ["ORC_LIB"]
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "kcount_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liborc.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_murmur3_x64_64.restype = C.c_uint64
        L.orc_murmur3_x64_64.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_quick_hash.restype = C.c_uint64
        L.orc_quick_hash.argtypes = [C.c_uint64]
        L.orc_num_longs.restype = C.c_int
        L.orc_num_longs.argtypes = [C.c_int]
        L.orc_minimizer_len.restype = C.c_int
        L.orc_minimizer_len.argtypes = [C.c_int]
        L.orc_pack_kmer.argtypes = [C.c_char_p, C.c_int, u64p]
        L.orc_get_kmers.restype = C.c_int
        L.orc_get_kmers.argtypes = [C.c_char_p, C.c_int, C.c_int, u64p]
        L.orc_revcomp.argtypes = [u64p, C.c_int, u64p]
        L.orc_kmer_less.restype = C.c_int
        L.orc_kmer_less.argtypes = [u64p, u64p, C.c_int]
        L.orc_kmer_hash.restype = C.c_uint64
        L.orc_kmer_hash.argtypes = [u64p, C.c_int]
        L.orc_minimizer.restype = C.c_uint64
        L.orc_minimizer.argtypes = [u64p, C.c_int, C.c_int]
        L.orc_minimizer_hash.restype = C.c_uint64
        L.orc_minimizer_hash.argtypes = [u64p, C.c_int, C.c_int]
        L.orc_target_rank.restype = C.c_int
        L.orc_target_rank.argtypes = [u64p, C.c_int, C.c_int, C.c_int]
        L.orc_get_ext.restype = C.c_char
        L.orc_get_ext.argtypes = [u16p, C.c_uint16, C.c_int]
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_add_reads.restype = C.c_int
        L.orc_add_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.orc_insert_supermer.restype = C.c_int
        L.orc_insert_supermer.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
        L.orc_add_ctg.restype = C.c_int
        L.orc_add_ctg.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        L.orc_build_supermers.restype = C.c_int
        L.orc_build_supermers.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_finalize.restype = C.c_int
        L.orc_finalize.argtypes = [C.c_void_p]
        L.orc_finalize_ex.restype = C.c_int
        L.orc_finalize_ex.argtypes = [C.c_void_p, C.c_int]
        L.orc_num_results.restype = C.c_uint64
        L.orc_num_results.argtypes = [C.c_void_p]
        L.orc_get_results.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_dump_table.restype = C.c_uint64
        L.orc_dump_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_get_stats.argtypes = [C.c_void_p, u64p]
        L.orc_kmer_to_string.argtypes = [u64p, C.c_int, C.c_char_p]
        _LIB = L
    return _LIB


def _u64(a):
    return a.ctypes.data_as(u64p)


def num_longs(k):
    return lib().orc_num_longs(k)


def pack_kmer(s, k=None):
    k = k or len(s)
    out = np.zeros(num_longs(k), dtype=np.uint64)
    lib().orc_pack_kmer(s.encode() if isinstance(s, str) else s, k, _u64(out))
    return out


def get_kmers(seq, k):
    b = seq.encode() if isinstance(seq, str) else seq
    n = max(0, len(b) - k + 1)
    out = np.zeros((n, num_longs(k)), dtype=np.uint64)
    if n:
        lib().orc_get_kmers(b, len(b), k, _u64(out))
    return out


def revcomp(words, k):
    w = np.ascontiguousarray(words, dtype=np.uint64)
    out = np.zeros_like(w)
    lib().orc_revcomp(_u64(w), k, _u64(out))
    return out


def kmer_hash(words):
    w = np.ascontiguousarray(words, dtype=np.uint64)
    return lib().orc_kmer_hash(_u64(w), len(w))


def minimizer_hash(words, k, m):
    w = np.ascontiguousarray(words, dtype=np.uint64)
    return lib().orc_minimizer_hash(_u64(w), k, m)


def target_rank(words, k, m, nranks):
    w = np.ascontiguousarray(words, dtype=np.uint64)
    return lib().orc_target_rank(_u64(w), k, m, nranks)


def get_ext(counts4, count, dmin_thres=2):
    c = np.ascontiguousarray(counts4, dtype=np.uint16)
    return lib().orc_get_ext(c.ctypes.data_as(u16p), int(count), dmin_thres).decode()


def kmer_to_string(words, k):
    w = np.ascontiguousarray(words, dtype=np.uint64)
    buf = C.create_string_buffer(k + 1)
    lib().orc_kmer_to_string(_u64(w), k, buf)
    return buf.value.decode()


STAT_NAMES = ("reads", "raw_kmers", "supermers", "kmers_inserted", "unique", "purged", "total_kmers",
              "sum_counts", "dropped", "nranks", "nthreads", "rehashes")


class Oracle:
    """End-to-end CPU kcount: add_reads()* -> finalize() -> results."""

    def __init__(self, k, qual_offset=33, dmin_thres=2, nranks=1, nthreads=1, capacity_per_rank=0):
        self.k, self.nl = k, num_longs(k)
        self._h = lib().orc_create(k, qual_offset, dmin_thres, nranks, nthreads, capacity_per_rank)
        if not self._h:
            raise ValueError("orc_create failed (k=%d)" % k)

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def add_reads(self, bases, quals, offsets, block_reads=200_000):
        """bases/quals: uint8 arrays of concatenated ASCII; offsets: uint64[nreads+1]."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        for r0 in range(0, n, block_reads):
            r1 = min(n, r0 + block_reads)
            off = offsets[r0:r1 + 1]
            rc = lib().orc_add_reads(self._h, bases.ctypes.data, quals.ctypes.data, off.ctypes.data, r1 - r0)
            if rc:
                raise RuntimeError("orc_add_reads failed: %d" % rc)

    def add_ctg(self, seq, depth):
        """The contig pass (after every read): process_seq(seq, depth) + insert_supermer_from_ctg (kcount_cpu.cpp:357-407)."""
        b = seq.encode() if isinstance(seq, str) else seq
        rc = lib().orc_add_ctg(self._h, b, len(b), int(depth))
        if rc:
            raise RuntimeError("orc_add_ctg failed: %d" % rc)

    def insert_supermer(self, target, seq):
        b = seq.encode() if isinstance(seq, str) else seq
        rc = lib().orc_insert_supermer(self._h, target, b, len(b))
        if rc:
            raise RuntimeError("orc_insert_supermer failed: %d" % rc)

    def supermers(self, masked_read):
        """SeqBlockInserter::process_seq on one case-masked read: [(target, start, length)] (kcount_cpu.cpp:73-103)."""
        b = masked_read.encode() if isinstance(masked_read, str) else masked_read
        n = max(len(b), 1)
        t = np.zeros(n, dtype=np.int32)
        s = np.zeros(n, dtype=np.int32)
        ln = np.zeros(n, dtype=np.int32)
        m = lib().orc_build_supermers(self._h, b, len(b), t.ctypes.data, s.ctypes.data, ln.ctypes.data)
        if m < 0:
            raise MemoryError
        return [(int(t[i]), int(s[i]), int(ln[i])) for i in range(m)]

    def finalize_unsorted(self):
        """Vote, purge and collect like the reference does (per rank, in parallel, no sort); returns the number of
        results.  What the CPU baseline times; the tests use finalize(), which also sorts by key."""
        if lib().orc_finalize_ex(self._h, 0):
            raise MemoryError
        return lib().orc_num_results(self._h)

    def finalize(self):
        if lib().orc_finalize(self._h):
            raise MemoryError
        n = lib().orc_num_results(self._h)
        keys = np.zeros((n, self.nl), dtype=np.uint64)
        counts = np.zeros(n, dtype=np.uint16)
        left = np.zeros(n, dtype=np.uint8)
        right = np.zeros(n, dtype=np.uint8)
        lib().orc_get_results(self._h, keys.ctypes.data, counts.ctypes.data, left.ctypes.data, right.ctypes.data)
        return keys, counts, left, right

    def dump_table(self):
        n = lib().orc_dump_table(self._h, None, None, None)
        keys = np.zeros((n, self.nl), dtype=np.uint64)
        counts = np.zeros(n, dtype=np.uint16)
        exts = np.zeros((n, 8), dtype=np.uint16)
        if n:
            lib().orc_dump_table(self._h, keys.ctypes.data, counts.ctypes.data, exts.ctypes.data)
        return keys, counts, exts

    def stats(self):
        s = np.zeros(len(STAT_NAMES), dtype=np.uint64)
        lib().orc_get_stats(self._h, _u64(s))
        return dict(zip(STAT_NAMES, (int(x) for x in s)))


def reads_to_arrays(reads, quals=None, qual_char="I"):
    """list of str -> (bases u8, quals u8, offsets u64)."""
    if quals is None:
        quals = [qual_char * len(r) for r in reads]
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    b = np.frombuffer("".join(reads).encode(), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
    q = np.frombuffer("".join(quals).encode(), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
    return b, q, offs


def count_reads(reads, quals=None, k=21, **kw):
    o = Oracle(k, **kw)
    b, q, offs = reads_to_arrays(reads, quals)
    o.add_reads(b, q, offs)
    res = o.finalize()
    st = o.stats()
    o.close()
    return res, st
