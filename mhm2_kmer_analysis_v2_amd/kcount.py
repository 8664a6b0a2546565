"""Python host side over the C ABI: mirrors the reference's analyze_kmers flow.

  reference (src/kcount/kcount.cpp:142-161)          here
  -------------------------------------------        --------------------------------
  KmerDHT ctor -> HashTableInserter::init             KmerCounter(k, ...)      kc_create
  count_kmers: per read quality-mask + process_seq    .submit_reads(...)       kc_submit_reads
  kmer_dht->flush_updates()                           .flush()                 kc_flush
  kmer_dht->finish_updates()                          .finalize()              kc_finalize
  local_kmers (KmerMap)                               .results()               kc_copy_results
  dump_kmers ("KMER count L R", kmer_dht.cpp:284)     .dump_lines()
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, kc_config, kc_kernel_time, kc_result, kc_stats, kc_synth_params, kc_tuning, lib


def _ptr(a):
    """numpy array / torch tensor / int / None -> address, and whether it is device memory."""
    if a is None:
        return None, False
    if isinstance(a, int):
        return a, True
    if hasattr(a, "data_ptr"):  # torch tensor
        return a.data_ptr(), bool(a.is_cuda)
    return a.ctypes.data, False


class _DeviceWords:
    """Raw device memory of the library as something torch can view without a copy (__cuda_array_interface__)."""

    def __init__(self, ptr, nwords, device):
        self.__cuda_array_interface__ = {"shape": (nwords,), "typestr": "<i8", "data": (ptr, False), "version": 2}
        self.device = device

    def tensor(self):
        import torch
        return torch.as_tensor(self, device=self.device)


def synth_params(**kw):
    p = kc_synth_params()
    lib().kc_synth_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError("unknown synth parameter %r" % k)
        setattr(p, k, v)
    return p


def synth_reads_host(nreads, first_read=0, params=None):
    """Host generator (same bytes as the device generator): bases u8, quals u8, offsets u64."""
    p = params or synth_params()
    L = p.read_len
    bases = np.empty(nreads * L, dtype=np.uint8)
    quals = np.empty(nreads * L, dtype=np.uint8)
    offs = np.empty(nreads + 1, dtype=np.uint64)
    check(lib().kc_synth_reads_host(C.byref(p), first_read, nreads, bases.ctypes.data, quals.ctypes.data, offs.ctypes.data),
          "kc_synth_reads_host")
    return bases, quals, offs


class KmerCounter:
    """One shard (one GPU) of the k-mer analysis stage."""

    def __init__(self, kmer_len, qual_offset=33, dmin_thres=2, device=0, rank_me=0, rank_n=1, max_elems=0, time_kernels=False,
                 max_kmers_buffered=0, tuning=None, reference_owner=False, shard_buckets=False, wire_units=False):
        L = lib()
        cfg = kc_config(kmer_len=kmer_len, qual_offset=qual_offset, dmin_thres=dmin_thres, device=device, rank_me=rank_me,
                        rank_n=rank_n, max_elems=max_elems, flags=(_lib.KC_FLAG_TIME_KERNELS if time_kernels else 0) | (_lib.KC_FLAG_REFERENCE_OWNER if reference_owner else 0) | (_lib.KC_FLAG_SHARD_BUCKETS if shard_buckets else 0) | (_lib.KC_FLAG_WIRE_UNITS if wire_units else 0),
                        reserved=0,
                        max_kmers_buffered=max_kmers_buffered)
        st = C.c_int(0)
        self._h = L.kc_create(C.byref(cfg), C.byref(st))
        if not self._h:
            raise _lib.KcError(st.value, "kc_create")
        self._wire_units = bool(wire_units)
        self.k = kmer_len
        self.nl = L.kc_num_longs(kmer_len)          # words of a k-mer in results, dumps and lookups (the reference's)
        self.rec_nl = L.kc_record_longs(kmer_len)   # words of a record on the shard wire (extract_partition / insert_records)
        self.rank_me, self.rank_n = rank_me, rank_n
        self.device = device
        self._tuning = tuning
        if tuning:
            self.set_tuning(**tuning)

    def set_tuning(self, **kw):
        """Geometry overrides of the bucketed path (tests / tuning): mode, writers, p1, p2, slots,
        chunk1, chunk2, chain1_max, chain2_max, arena1, ovf_capacity."""
        t = kc_tuning()
        for k, v in kw.items():
            if not hasattr(t, k):
                raise TypeError("unknown tuning field %r" % k)
            setattr(t, k, v)
        check(lib().kc_set_tuning(self._h, C.byref(t)), "kc_set_tuning")

    def close(self):
        if getattr(self, "_h", None):
            lib().kc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, stream_handle):
        check(lib().kc_set_stream(self._h, stream_handle), "kc_set_stream")

    def reset(self, new_kmer_len=0):
        check(lib().kc_reset(self._h, new_kmer_len), "kc_reset")
        if self._tuning:
            self.set_tuning(**self._tuning)
        if new_kmer_len:
            self.k = new_kmer_len
            self.nl = lib().kc_num_longs(new_kmer_len)
            self.rec_nl = lib().kc_record_longs(new_kmer_len)

    def submit_reads(self, bases, quals, offsets, nreads=None):
        pb, dev = _ptr(bases)
        pq, _ = _ptr(quals)
        po, _ = _ptr(offsets)
        n = (len(offsets) - 1) if nreads is None else nreads
        check(lib().kc_submit_reads(self._h, pb, pq, po, n, 1 if dev else 0), "kc_submit_reads")

    def submit_packed_reads(self, packed, offsets, nreads=None):
        """PackedRead bytes (base | quality<<3, src/packed_reads.cpp:99-126) + offsets."""
        pp, dev = _ptr(packed)
        po, _ = _ptr(offsets)
        n = (len(offsets) - 1) if nreads is None else nreads
        check(lib().kc_submit_packed_reads(self._h, pp, po, n, 1 if dev else 0), "kc_submit_packed_reads")

    def submit_seq_block(self, seqs, length=None):
        if isinstance(seqs, (bytes, bytearray)):
            buf = np.frombuffer(bytes(seqs), dtype=np.uint8)
            check(lib().kc_submit_seq_block(self._h, buf.ctypes.data, len(buf), 0), "kc_submit_seq_block")
            return
        p, dev = _ptr(seqs)
        n = len(seqs) if length is None else length
        check(lib().kc_submit_seq_block(self._h, p, n, 1 if dev else 0), "kc_submit_seq_block")

    def wire_unit(self):
        """(words, records, pieces) of kc_extract_partition / kc_insert_records: a unit of `words` words holds `records`
        records, every destination gets `pieces` pieces -- (kc_record_longs, 1, 1) for k-mer records, (3, 4, 2..16) where a
        context created with wire_units=True exchanges six-byte records (kc_wire_unit)."""
        w, r, q = C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().kc_wire_unit(self._h, C.byref(w), C.byref(r), C.byref(q)), "kc_wire_unit")
        return w.value, r.value, q.value

    def partition_owner(self, kmer_words):
        """the shard kc_extract_partition sends this canonical k-mer to (kc_partition_owner)"""
        w = np.ascontiguousarray(kmer_words, dtype=np.uint64)
        o = C.c_int(-1)
        check(lib().kc_partition_owner(self._h, w.ctypes.data, C.byref(o)), "kc_partition_owner")
        return o.value

    def extract_partition(self, bases, quals, offsets, records, seg_capacity, nreads=None):
        """records: device buffer of rank_n*pieces*seg_capacity*unit_words u64 (wire_unit()).  Returns the units of every
        piece, destination after destination."""
        pb, dev = _ptr(bases)
        pq, _ = _ptr(quals)
        po, _ = _ptr(offsets)
        pr, _ = _ptr(records)
        n = (len(offsets) - 1) if nreads is None else nreads
        counts = np.zeros(self.rank_n * (self.wire_unit()[2] if self._wire_units else 1), dtype=np.uint64)
        check(lib().kc_extract_partition(self._h, pb, pq, po, n, 1 if dev else 0, pr, seg_capacity, counts.ctypes.data),
              "kc_extract_partition")
        return counts

    def build_supermers(self, block, capacity=None):
        """ParseAndPackGPUDriver::process_seq_block + pack_seq_block on a '_'-joined case-masked block (bytes):
        returns (targets i32, offsets i32, lens u16, num_valid_kmers, packed u8[(len+1)//2])."""
        blk = np.frombuffer(block, dtype=np.uint8) if isinstance(block, (bytes, bytearray)) else np.ascontiguousarray(block, dtype=np.uint8)
        cap = capacity if capacity is not None else max(16, len(blk))
        out = np.zeros(cap, dtype=np.dtype([("target", np.int32), ("offset", np.int32), ("len", np.uint16), ("pad", np.uint16)]))
        packed = np.zeros((len(blk) + 1) // 2, dtype=np.uint8)
        n, nk = C.c_uint32(0), C.c_uint32(0)
        check(lib().kc_build_supermers(self._h, blk.ctypes.data, len(blk), 0, out.ctypes.data, cap, C.byref(n), C.byref(nk),
                                       packed.ctypes.data), "kc_build_supermers")
        out = out[:n.value]
        return out["target"].copy(), out["offset"].copy(), out["len"].copy(), nk.value, packed

    def begin_ctg_kmers(self, max_ctg_kmers):
        """HashTableInserter::init_ctg_kmers: room for that many distinct contig k-mers (before finalize)."""
        check(lib().kc_begin_ctg_kmers(self._h, int(max_ctg_kmers)), "kc_begin_ctg_kmers")

    def submit_ctgs(self, ctgs, depths):
        """process_seq(ctg.seq, ctg.depth) for every contig: a '_'-joined block with its per-character depths."""
        block = b"_".join(c.encode() if isinstance(c, str) else c for c in ctgs) + b"_"
        dd = np.zeros(len(block), dtype=np.uint16)
        at = 0
        for c, d in zip(ctgs, depths):
            dd[at:at + len(c) + 1] = d
            at += len(c) + 1
        blk = np.frombuffer(block, dtype=np.uint8)
        check(lib().kc_submit_ctg_block(self._h, blk.ctypes.data, dd.ctypes.data, len(blk), 0), "kc_submit_ctg_block")

    def ctg_stats(self):
        """(distinct contig k-mers in the contig table, characters submitted)"""
        d, n = C.c_uint64(0), C.c_uint64(0)
        check(lib().kc_ctg_stats(self._h, C.byref(d), C.byref(n)), "kc_ctg_stats")
        return int(d.value), int(n.value)

    def submit_packed_supermers(self, packed):
        """4-bit packed supermers joined by the byte '_' (HashTableGPUDriver::insert_supermer's buffer)."""
        pp, dev = _ptr(packed)
        n = packed.numel() if hasattr(packed, "numel") else len(packed)
        check(lib().kc_submit_packed_supermers(self._h, pp, n, 1 if dev else 0), "kc_submit_packed_supermers")

    def insert_records(self, records, n):
        pr, _ = _ptr(records)
        check(lib().kc_insert_records(self._h, pr, n), "kc_insert_records")

    def insert_record_pieces(self, records, piece_stride_units, units):
        """kc_insert_record_pieces: pieces of `units[j]` units that lie piece_stride_units units apart, from `records` on"""
        pr, _ = _ptr(records)
        u = np.ascontiguousarray(units, dtype=np.uint64)
        check(lib().kc_insert_record_pieces(self._h, pr, int(piece_stride_units), len(u), u.ctypes.data), "kc_insert_record_pieces")

    # ---- the single-pass shard flow (ownership by level-1 bucket) ----
    def shard_extract(self, bases, quals, offsets, segments, seg_words, nreads=None):
        """segments: device buffer of rank_n*seg_words u64.  Returns the words to ship per destination."""
        pb, dev = _ptr(bases)
        pq, _ = _ptr(quals)
        po, _ = _ptr(offsets)
        ps, _ = _ptr(segments)
        n = (len(offsets) - 1) if nreads is None else nreads
        words = np.zeros(self.rank_n, dtype=np.uint64)
        check(lib().kc_shard_extract(self._h, pb, pq, po, n, 1 if dev else 0, ps, seg_words, words.ctypes.data), "kc_shard_extract")
        return words

    def shard_extract_seq_block(self, seqs, segments, seg_words):
        """The same from a '_'-joined case-masked block (bytes, or a device tensor of them)."""
        if isinstance(seqs, (bytes, bytearray)):
            buf = np.frombuffer(bytes(seqs), dtype=np.uint8)
            p, dev, n = buf.ctypes.data, False, len(buf)
        else:
            (p, dev), n = _ptr(seqs), len(seqs)
        ps, _ = _ptr(segments)
        words = np.zeros(self.rank_n, dtype=np.uint64)
        check(lib().kc_shard_extract_seq_block(self._h, p, n, 1 if dev else 0, ps, seg_words, words.ctypes.data), "kc_shard_extract_seq_block")
        return words

    def shard_reserve(self, nwords, device=None):
        """Context-owned device memory for nwords incoming u64, as an int64 torch tensor viewing it (no copy)."""
        import torch
        p = C.c_void_p(0)
        check(lib().kc_shard_reserve(self._h, int(nwords), C.byref(p)), "kc_shard_reserve")
        if not nwords:
            return torch.empty(0, dtype=torch.int64, device=device or ("cuda:%d" % self.device))
        return _DeviceWords(p.value, int(nwords), device or ("cuda:%d" % self.device)).tensor()

    def shard_commit(self, segment, nwords):
        ps, _ = _ptr(segment)
        check(lib().kc_shard_commit(self._h, ps, int(nwords)), "kc_shard_commit")

    def shard_capacity(self):
        """Distinct k-mers this shard's regions hold in the single-pass flow (kc_shard_capacity)."""
        v = C.c_uint64(0)
        check(lib().kc_shard_capacity(self._h, C.byref(v)), "kc_shard_capacity")
        return v.value

    def shard_owner(self, kmer_words):
        w = np.ascontiguousarray(kmer_words, dtype=np.uint64)
        o = C.c_int(-1)
        check(lib().kc_shard_owner(self._h, w.ctypes.data, C.byref(o)), "kc_shard_owner")
        return o.value

    def flush(self):
        check(lib().kc_flush(self._h), "kc_flush")

    def finalize(self):
        r = kc_result()
        check(lib().kc_finalize(self._h, C.byref(r)), "kc_finalize")
        self._res = r
        return r

    def results(self):
        """Host copies: keys (n, num_longs) u64, counts u16, left u8, right u8 (unordered)."""
        r = self.finalize()
        n = int(r.n)
        keys = np.empty((n, self.nl), dtype=np.uint64)
        counts = np.empty(n, dtype=np.uint16)
        left = np.empty(n, dtype=np.uint8)
        right = np.empty(n, dtype=np.uint8)
        check(lib().kc_copy_results(self._h, keys.ctypes.data, counts.ctypes.data, left.ctypes.data, right.ctypes.data),
              "kc_copy_results")
        return keys, counts, left, right

    def lookup(self, queries):
        """queries: (n, num_longs) uint64 k-mers in either orientation (host array).  Returns counts u16, left u8,
        right u8; count 0 = not among the results (KmerDHT::get_kmer_count, kmer_dht.cpp:228-245)."""
        self.finalize()
        qa = np.ascontiguousarray(queries, dtype=np.uint64).reshape(-1, self.nl)
        n = len(qa)
        counts = np.zeros(n, dtype=np.uint16)
        left = np.zeros(n, dtype=np.uint8)
        right = np.zeros(n, dtype=np.uint8)
        check(lib().kc_lookup(self._h, qa.ctypes.data, n, 0, counts.ctypes.data, left.ctypes.data, right.ctypes.data), "kc_lookup")
        return counts, left, right

    def sorted_results(self):
        keys, counts, left, right = self.results()
        order = np.lexsort([keys[:, j] for j in range(self.nl - 1, -1, -1)]) if len(counts) else np.zeros(0, dtype=np.int64)
        return keys[order], counts[order], left[order], right[order]

    def dump_table(self):
        n = C.c_uint64(0)
        check(lib().kc_dump_table(self._h, None, None, None, C.byref(n)), "kc_dump_table")
        keys = np.empty((n.value, self.nl), dtype=np.uint64)
        counts = np.empty(n.value, dtype=np.uint16)
        exts = np.empty((n.value, 8), dtype=np.uint16)
        if n.value:
            check(lib().kc_dump_table(self._h, keys.ctypes.data, counts.ctypes.data, exts.ctypes.data, C.byref(n)), "kc_dump_table")
        order = np.lexsort([keys[:, j] for j in range(self.nl - 1, -1, -1)]) if n.value else np.zeros(0, dtype=np.int64)
        return keys[order], counts[order], exts[order]

    def stats(self):
        s = kc_stats()
        check(lib().kc_get_stats(self._h, C.byref(s)), "kc_get_stats")
        return {n: int(getattr(s, n)) for n, _ in kc_stats._fields_}

    def arena_probe_rate(self):
        """TB/s of level 1's write pattern on the level-1 arena this context chose (0.0: no probe ran)."""
        v = C.c_double(0.0)
        check(lib().kc_arena_probe_rate(self._h, C.byref(v)), "kc_arena_probe_rate")
        return float(v.value)

    def kernel_times(self, clear=False):
        """{kernel name: (launches, total_ms)} from HIP events on the launch stream (needs time_kernels=True)."""
        arr = (kc_kernel_time * 32)()
        n = C.c_int(0)
        check(lib().kc_get_kernel_times(self._h, arr, 32, C.byref(n)), "kc_get_kernel_times")
        out = {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].total_ms)) for i in range(min(n.value, 32))}
        if clear:
            check(lib().kc_clear_kernel_times(self._h), "kc_clear_kernel_times")
        return out

    def synth_reads_device(self, d_bases, d_quals, d_offsets, nreads, first_read=0, params=None):
        p = params or synth_params()
        check(lib().kc_synth_reads_device(self._h, C.byref(p), first_read, nreads, _ptr(d_bases)[0], _ptr(d_quals)[0],
                                          _ptr(d_offsets)[0]), "kc_synth_reads_device")

    def dump_lines(self):
        """The reference's dump format, one "KMER count L R" per k-mer (kmer_dht.cpp:273-297), sorted."""
        keys, counts, left, right = self.sorted_results()
        return ["%s %d %s %s" % (kmer_to_string(keys[i], self.k), counts[i], chr(left[i]), chr(right[i]))
                for i in range(len(counts))]


    def dump_kmers(self, directory=".", rank=None):
        """KmerDHT::dump_kmers (src/kcount/kmer_dht.cpp:273-297): per-rank gzip text file "kmers-<k>.txt.gz", one
        "KMER count L R" line per k-mer.  Returns the path."""
        import gzip
        import os
        r = self.rank_me if rank is None else rank
        d = os.path.join(directory, "rank_%d" % r) if self.rank_n > 1 else directory
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "kmers-%d.txt.gz" % self.k)
        with gzip.open(path, "wt") as f:
            for line in self.dump_lines():
                f.write(line + "\n")
        return path


def kmer_to_string(words, k):
    return "".join("ACGT"[(int(words[i // 32]) >> (2 * (31 - (i % 32)))) & 3] for i in range(k))


def fastq_to_packed(text, qual_offset=33):
    """FASTQ text (bytes) -> (packed u8, offsets u64): the read cache's bytes of src/packed_reads.cpp:99-126, ready for
    KmerCounter.submit_packed_reads.  Host only (FastqReader::get_next_fq_record's unpaired pass, src/fastq.cpp:1028+)."""
    data = text.encode() if isinstance(text, str) else bytes(text)
    n, nb = C.c_uint64(0), C.c_uint64(0)
    st = lib().kc_fastq_to_packed(data, len(data), qual_offset, None, 0, None, 0, C.byref(n), C.byref(nb))
    if st not in (_lib.KC_OK, _lib.KC_ERR_CAPACITY):
        check(st, "kc_fastq_to_packed")
    packed = np.zeros(max(nb.value, 1), dtype=np.uint8)
    offs = np.zeros(n.value + 1, dtype=np.uint64)
    check(lib().kc_fastq_to_packed(data, len(data), qual_offset, packed.ctypes.data, nb.value, offs.ctypes.data, n.value, C.byref(n),
                                   C.byref(nb)), "kc_fastq_to_packed")
    return packed[:nb.value], offs


def analyze_kmers(kmer_len, qual_offset, bases, quals, offsets, dmin_thres=2, device=0, max_elems=0, tuning=None):
    """analyze_kmers (src/kcount/kcount.cpp:142-161) for one shard: returns sorted results and stats."""
    with KmerCounter(kmer_len, qual_offset, dmin_thres, device=device, max_elems=max_elems, tuning=tuning) as kc:
        kc.submit_reads(bases, quals, offsets)
        kc.flush()
        res = kc.sorted_results()
        return res, kc.stats()
