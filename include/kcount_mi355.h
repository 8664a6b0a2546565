/*
 * kcount_mi355.h -- C ABI of libkcount_mi355.so, the MI355X (gfx950) k-mer
 * analysis stage that drops in behind MHM2's kcount / KmerDHT operator surface.
 *
 * Plain C: opaque handle, plain pointers and sizes, int status codes, no
 * exceptions, no STL, no torch types.  Every entry point names the reference
 * interface it replaces (paths relative to the reference checkout).  The C++
 * adapters that re-create the reference's own driver classes on top of this
 * ABI are in mhm2_kmer_analysis_v2_amd/csrc/kcount_driver.hpp; the binding a
 * reference maintainer would add is shown in INTEGRATION.md.
 *
 * Semantics are those of the reference *CPU* backend (src/kcount/kcount_cpu.cpp),
 * spec S1-S9 in SURVEY.md section 8a -- not the reference GPU backend, which
 * differs from it (N handling, counter saturation).
 */
#ifndef KCOUNT_MI355_H
#define KCOUNT_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KC_ABI_VERSION 1

/* status codes (all entry points returning int) */
enum {
  KC_OK = 0,
  KC_ERR_INVALID_ARG = -1,
  KC_ERR_UNSUPPORTED_K = -2, /* k < 3 or k > 125 (126 and 127 would need a fifth record word) */
  KC_ERR_NO_DEVICE = -3,     /* HIP runtime reports no usable gfx950 device */
  KC_ERR_HIP = -4,           /* a HIP call failed; kc_last_error() has the text */
  KC_ERR_OUT_OF_MEMORY = -5,
  KC_ERR_CAPACITY = -6,      /* a buffer is too small: a caller-provided segment (retry with more room), or the k-mer buffer /
                                its overflow lists (raise max_kmers_buffered); kc_last_error says which */
  KC_ERR_BAD_BASE = -7,      /* a read holds a byte outside ACGTN/acgtn (reference: DIE at kcount_cpu.cpp:484-486) */
  KC_ERR_STATE = -8          /* call not allowed in this state (e.g. submit after finalize without reset) */
};

typedef struct kc_ctx kc_ctx;

/*
 * Run-time configuration.  Replaces the compile-time / CLI knobs the reference
 * spreads over CMakeDefinitions.txt and src/options.cpp:
 *   kmer_len     Options::kmer_lens                       src/options.hpp:80
 *   qual_offset  Options::qual_offset, SeqBlockInserter   src/kcount/kcount.hpp:60
 *   dmin_thres   --min-depth-thres, _dmin_thres           src/kcount/kmer_dht.hpp:57, src/kcount/kcount.cpp:146
 *   rank_me/n    upcxx::rank_me()/rank_n() as passed to
 *                HashTableGPUDriver::init                 src/kcount/kcount-gpu/gpu_hash_table.hpp:155
 *   max_elems    init(max_elems, ..., num_errors, ...)    same; here: expected distinct k-mers of this shard
 *                (0 = pick a default; the table grows instead of dropping inserts)
 */
typedef struct kc_config {
  int32_t kmer_len;
  int32_t qual_offset; /* 33 or 64 */
  int32_t dmin_thres;  /* default 2 */
  int32_t device;      /* HIP device ordinal */
  int32_t rank_me;     /* this shard owns k-mers with kc_owner(key) == rank_me */
  int32_t rank_n;      /* shards in the job (GPUs); 1 = everything local */
  uint64_t max_elems;
  uint32_t flags; /* KC_FLAG_* */
  uint32_t reserved;
  /* k-mer occurrences the context buffers on its fast (bucketed) path.  Sized for the whole input of a pass
   * (kc_reset .. kc_finalize) the stage runs in one pass, which is what the benchmark measures.  When more arrives,
   * what is buffered is counted, merged into the global table (one table operation per distinct k-mer) and the
   * buffer starts again empty: the reference streams 1 MB insert blocks into its one table for as long as reads come
   * (gpu_hash_table.cpp:681-695).  Correct for any size, but several times slower than one pass (the table then holds
   * every distinct k-mer, singletons included).  Role of the reference's my_num_kmers estimate
   * (src/contigging.cpp:86).  0 = 64 Mi. */
  uint64_t max_kmers_buffered;
} kc_config;

#define KC_FLAG_NONE 0u
#define KC_FLAG_REFERENCE_OWNER 2u /* owner shard = the reference's KmerDHT::get_kmer_target_rank (quick_hash of the minimizer,
                                     src/kcount/kmer_dht.cpp:117-119,192-196) instead of the k-mer hash: for runs mixed with
                                     unmodified MHM2 ranks; slower (k-m+1 m-mer comparisons per k-mer) */
#define KC_FLAG_SHARD_BUCKETS 4u /* this context will run the single-pass shard flow (kc_shard_*): its regions are sized for 1/rank_n
                                 * of the level-1 buckets holding all of max_elems (see kc_shard_capacity) */
#define KC_FLAG_WIRE_UNITS 8u /* kc_extract_partition / kc_insert_records exchange UNITS of the library's own wire record where the
                                 geometry has one (kc_wire_unit: four six-byte records of a mixed k-mer per three words for k = 21;
                                 one k-mer record otherwise) and the owner of a k-mer is kc_partition_owner, not kc_owner.  All
                                 shards of an exchange must be created alike. */
#define KC_FLAG_TIME_KERNELS 1u /* bracket every kernel launch with HIP events on its own stream (kc_get_kernel_times) */

/* Scalars the reference logs (src/kcount/kcount.cpp:94-102,158-160;
 * src/kcount/kcount_cpu.cpp:495-521,586-598) plus table geometry. */
typedef struct kc_stats {
  uint64_t num_reads;
  uint64_t num_bases;
  uint64_t raw_kmers;      /* sum over reads of max(0, len-k+1): kcount.cpp:86 -- the unit of the k-mers/s metric */
  uint64_t kmers_inserted; /* k-mer occurrences with both neighbours (S5) put into this shard's table */
  uint64_t num_unique;     /* table entries before the purge */
  uint64_t num_purged;     /* entries removed by S8 */
  uint64_t total_kmers;    /* results: "Total kmers" */
  uint64_t sum_counts;     /* "Total kmer count sum" */
  uint64_t num_dropped;    /* always 0: the table grows; kept for the reference's precondition check */
  uint64_t capacity;       /* table slots */
  uint64_t num_gpu_calls;  /* kernel launches so far (HashTableGPUDriver::get_num_gpu_calls) */
  uint64_t table_bytes;
} kc_stats;

/* Dense result arrays resident in HBM (replaces the compact KmerExtsMap the
 * reference copies back slot by slot, gpu_hash_table.cpp:205-245,776-827, and
 * the KmerCounts it becomes, kmer_dht.hpp:62-68).  Unordered, like a hash-map
 * iteration; entry i is keys[i*num_longs .. +num_longs), counts[i], left[i],
 * right[i] with left/right in "ACGT".  Pointers stay valid until kc_reset /
 * kc_destroy. */
typedef struct kc_result {
  uint64_t n;
  int32_t num_longs;
  int32_t reserved;
  const uint64_t *d_keys;
  const uint16_t *d_counts;
  const uint8_t *d_left;
  const uint8_t *d_right;
} kc_result;

/* ---- library ------------------------------------------------------------ */
int kc_abi_version(void);
const char *kc_error_string(int status);
const char *kc_last_error(void); /* text of the last failing HIP call on this thread */
int kc_device_count(void);       /* 0 without a GPU; never aborts */

/* Kmer<MAX_K>::N_LONGS for the MAX_K the reference would pick: k/32+1 (src/main.cpp:169-190, src/kmer.hpp:64). */
int kc_num_longs(int kmer_len);
/* Words of one k-mer RECORD on the shard wire (kc_extract_partition / kc_insert_records): kc_num_longs, plus one when
 * k % 32 is 30 or 31 (the last key word then has no six spare bits for the two extension codes, which ride in a word of
 * their own).  Results, dumps and lookups always use kc_num_longs. */
int kc_record_longs(int kmer_len);
/* Shard that owns a canonical k-mer (role of KmerDHT::get_kmer_target_rank,
 * src/kcount/kmer_dht.cpp:192-196; any deterministic function of the k-mer
 * gives the same final set).  Host-callable. */
int kc_owner(const uint64_t *kmer_words, int kmer_len, int rank_n);
/* The reference's own target rank, bit for bit: quick_hash(minimizer(kmer, m)) % rank_n with m = clamp(2k/3+1, 15, 27)
 * (src/kmer.cpp:349-398,459-468, src/hash_funcs.c:332-342, src/kcount/kmer_dht.cpp:117-119,192-196).  Host-callable;
 * contexts created with KC_FLAG_REFERENCE_OWNER use it on the device. */
int kc_owner_reference(const uint64_t *kmer_words, int kmer_len, int rank_n);

/* ---- context ------------------------------------------------------------ */
/* HashTableGPUDriver::init + ParseAndPackGPUDriver ctor (gpu_hash_table.cpp:522-624, parse_and_pack.cpp:239-267). */
kc_ctx *kc_create(const kc_config *cfg, int *status);
/* ~HashTableGPUDriver / ~ParseAndPackGPUDriver */
void kc_destroy(kc_ctx *ctx);
/* Launch everything on this hipStream_t (e.g. torch's current stream).  NULL = the context's own stream. */
int kc_set_stream(kc_ctx *ctx, void *hip_stream);
/* Empty the table and results but keep every allocation (multi-k sweeps with a new
 * kmer_len re-use the arena; the reference re-allocates per run, F5 in SURVEY.md). */
int kc_reset(kc_ctx *ctx, int new_kmer_len);

/* ---- the hot path ------------------------------------------------------- */
/*
 * count_kmers' read loop + SeqBlockInserter::process_seq + the whole insert path
 * for reads whose k-mers this shard owns (src/kcount/kcount.cpp:71-90,
 * kcount_cpu.cpp:73-103,338-355).  bases/quals: concatenated ASCII, read r is
 * [offsets[r], offsets[r+1]); quality mask S2 is applied on the device.
 * on_device != 0: all three pointers are device pointers (HBM-resident input).
 * With rank_n > 1 only k-mers owned by rank_me are inserted (use
 * kc_extract_partition + kc_insert_records for the sharded flow).
 */
int kc_submit_reads(kc_ctx *ctx, const uint8_t *bases, const uint8_t *quals, const uint64_t *offsets, uint64_t nreads,
                    int on_device);

/*
 * The same from the reference's in-memory read cache, fed to the device as it is stored: one byte per base,
 * low 3 bits = A0 C1 G2 T3 N4, high 5 bits = min(quality - qual_offset, 31) (PackedRead,
 * src/packed_reads.cpp:99-126), read r = packed[offsets[r], offsets[r+1]).  Saves count_kmers the per-read
 * unpack into three strings (src/kcount/kcount.cpp:76-85, packed_reads.cpp:196-208) and half the input bytes.
 */
int kc_submit_packed_reads(kc_ctx *ctx, const uint8_t *packed, const uint64_t *offsets, uint64_t nreads, int on_device);

/*
 * FASTQ text -> the read cache's packed bytes, on the host (no context, no GPU): the unpaired pass of
 * FastqReader::get_next_fq_record (src/fastq.cpp:1028-1140: four lines per record, '@' name, sequence, '+' line,
 * qualities of the sequence's length; trailing white space and CR stripped) followed by PackedRead's constructor
 * (src/packed_reads.cpp:99-126: ACGT 0-3, N and the IUPAC codes 4, quality min(q - qual_offset, 31) << 3).  The reads
 * are appended to packed[0..packed_capacity) with offsets[r] .. offsets[r+1] (offsets[0] = 0, reads_capacity + 1 entries),
 * ready for kc_submit_packed_reads.  *nreads / *nbytes receive the totals of the whole text even when the arrays are too
 * small (KC_ERR_CAPACITY: call again with that much room; call with NULL arrays to size).  KC_ERR_BAD_BASE: a character
 * the reference DIEs on; KC_ERR_INVALID_ARG: a malformed record (kc_last_error names the line).
 * The reference's dummy mate of an unpaired read (the one-base read "N", src/merge_reads.cpp:372-377) holds no k-mer
 * and is not produced.
 */
int kc_fastq_to_packed(const char *text, uint64_t len, int qual_offset, uint8_t *packed, uint64_t packed_capacity, uint64_t *offsets,
                       uint64_t reads_capacity, uint64_t *nreads, uint64_t *nbytes);

/*
 * ParseAndPackGPUDriver::process_seq_block input format
 * (src/kcount/kcount_gpu.cpp:167-180, parse_and_pack.cpp:281-319): reads already
 * case-masked (lowercase = low quality) and joined by '_'.
 */
int kc_submit_seq_block(kc_ctx *ctx, const char *seqs, uint64_t len, int on_device);

/*
 * Sharded flow, sender side: extract k-mer records from a block of reads and
 * bin them by owner shard (replaces parse_and_pack + build_supermers + the
 * per-supermer ThreeTierAggrStore::update of kmer_dht.cpp:247-250).  Records of
 * shard d land in d_records[d*seg_capacity*L ...] with L = kc_record_longs(k); h_counts[d] receives
 * how many.  A record is L words: the canonical k-mer with the two
 * extension codes in the low 6 bits of its last word (left | right<<3; 0-3 =
 * ACGT, 4 = none).  KC_ERR_CAPACITY if a segment would overflow (nothing is lost:
 * the table is untouched, call again with more room).
 * Contexts created with KC_FLAG_WIRE_UNITS: records, counts and capacities are UNITS in PIECES, and h_counts has an entry
 * per piece -- see kc_wire_unit below.
 */
int kc_extract_partition(kc_ctx *ctx, const uint8_t *bases, const uint8_t *quals, const uint64_t *offsets,
                         uint64_t nreads, int on_device, uint64_t *d_records, uint64_t seg_capacity,
                         uint64_t *h_counts);
/* The same for a '_'-joined, case-masked block (the format ParseAndPackGPUDriver::process_seq_block takes). */
int kc_extract_partition_seq_block(kc_ctx *ctx, const char *seqs, uint64_t len, int on_device, uint64_t *d_records,
                                   uint64_t seg_capacity, uint64_t *h_counts);
/* Receiver side: HashTableGPUDriver::insert_supermer/insert_supermer_block
 * (gpu_hash_table.cpp:655-695) for records that arrived from other shards. */
int kc_insert_records(kc_ctx *ctx, const uint64_t *d_records, uint64_t n);

/* What kc_extract_partition writes and kc_insert_records reads, for contexts created with KC_FLAG_WIRE_UNITS: a unit is
 * *unit_words words holding *unit_records records, and every destination shard gets *pieces pieces: h_counts has
 * rank_n * *pieces entries, piece j = d * *pieces + q of destination d starts at d_records + j * seg_capacity *
 * *unit_words and holds h_counts[j] units; seg_capacity counts the units of ONE piece (a block of R reads of length L
 * needs about R * (L - k - 1) / (rank_n * *pieces * *unit_records) * 1.3 + 4096); kc_insert_records takes units, a piece
 * or any number of pieces laid end to end, in any order.  Where level 1 writes six-byte records (k = 21 with 1024
 * level-1 buckets) a unit is three words = four records of six bytes -- the k-mer travels mixed, as level 1 stages it,
 * so the receiver neither unpacks nor hashes it; a piece is closed to whole units with marker slots -- and the pieces
 * of a destination hold its records by the top bits of their level-1 bucket (2 to 16 pieces, 128 per sender at most),
 * which is what makes the receiver's level 1 cheap: a round of its 1024-way split that reads out of one piece meets a
 * sixteenth of the buckets and appends sixteen times as much to each.  Otherwise (and without the flag) a unit is one
 * k-mer record of kc_record_longs words and a destination has one piece.  Units are opaque. */
int kc_wire_unit(kc_ctx *ctx, int *unit_words, int *unit_records, int *pieces);
/* kc_insert_records for several pieces that lie `piece_stride_units` units apart (piece j at d_records + j *
 * piece_stride_units * unit_words, h_units[j] units; empty ones are skipped): what a shard keeps for itself of a block it
 * has extracted -- its own pieces of the send buffer, where they lie.  With wire units one launch takes up to sixteen
 * pieces (a round of a workgroup reads out of one), so that many small pieces do not become many small launches. */
int kc_insert_record_pieces(kc_ctx *ctx, const uint64_t *d_records, uint64_t piece_stride_units, int npieces, const uint64_t *h_units);
/* The shard kc_extract_partition sends a canonical k-mer to (role of KmerDHT::get_kmer_target_rank, kmer_dht.cpp:192-196):
 * kc_owner / kc_owner_reference for k-mer records; eight bits of the mixed k-mer for six-byte wire records (bits only
 * the probe stride of a region table uses, so every shard keeps the whole geometry). */
int kc_partition_owner(kc_ctx *ctx, const uint64_t *kmer_words, int *owner);

/* ---- the single-pass shard flow: a shard owns level-1 BUCKETS --------------------------------------------
 * Same role as kc_extract_partition + kc_insert_records -- the aggregated supermer exchange of
 * ThreeTierAggrStore<Supermer>::update / KmerDHT::flush_updates (src/kcount/kmer_dht.cpp:143-151,247-258) plus the
 * owner's insert_supermer_block (gpu_hash_table.cpp:655-695) -- without their two extra passes over the records: the
 * owner of a k-mer is the owner of the level-1 bucket it falls into (each shard a contiguous range of buckets;
 * kc_shard_owner), so the sender's ordinary level-1 pass has already sorted its records by destination.  What other
 * shards own is copied once into one wire segment per destination; what this shard owns never moves; a received segment
 * is read in place by the owner's level 2.  All shards of an exchange must be created with the same kmer_len,
 * max_kmers_buffered, max_elems, rank_n and tuning (a segment carries a signature; a mismatch is KC_ERR_INVALID_ARG).
 * A pass uses either this flow or the hash-ownership entry points (kc_submit_*, kc_insert_records), not both
 * (KC_ERR_STATE); a context on the global-table path (tuning mode 1, or out of buffer) has only the latter.
 */
/* Sender: count_kmers' loop body for one block of reads (src/kcount/kcount.cpp:71-90) + add_supermer for all of it.
 * d_segments: device buffer of rank_n * seg_words u64; segment d (at d * seg_words) receives what shard d owns,
 * h_words[d] = how many words of it to ship (0 for rank_me and for an empty block).  A block of R reads of length L
 * needs about R * (L - k - 1) / rank_n * kc_record_longs(k) * 1.1 + 1024 words per segment at most -- the compact records
 * of k <= 21 with 1024 level-1 buckets travel as FIVE BYTES each (the segment is sorted by bucket, which is therefore
 * implied: the reference compresses its wire too, as supermers, kmer_dht.cpp:69-100), 5/8 of that; h_words says what the
 * segment really holds.  A segment is opaque to the caller: ship h_words[d] words as they are.  KC_ERR_CAPACITY when a segment is
 * too small (the block's k-mers then stay buffered in this context; nothing was shipped) or when the context would
 * hold more than max_kmers_buffered. */
int kc_shard_extract(kc_ctx *ctx, const uint8_t *bases, const uint8_t *quals, const uint64_t *offsets, uint64_t nreads, int on_device,
                     uint64_t *d_segments, uint64_t seg_words, uint64_t *h_words);
/* The same for a '_'-joined, case-masked block (the format ParseAndPackGPUDriver::process_seq_block takes). */
int kc_shard_extract_seq_block(kc_ctx *ctx, const char *seqs, uint64_t len, int on_device, uint64_t *d_segments, uint64_t seg_words,
                               uint64_t *h_words);
/* Receiver: device memory of the context for `nwords` incoming words (the sum over the senders of one block; the caller
 * receives each sender's segment into its own 2-word-aligned part of it).  It belongs to the context and stays valid
 * until kc_reset / kc_destroy; earlier reservations never move. */
int kc_shard_reserve(kc_ctx *ctx, uint64_t nwords, uint64_t **d_dst);
/* Receiver: a whole segment from one sender has landed at d_segment (inside a kc_shard_reserve area, ordered before
 * the context's stream).  It becomes part of this shard's buckets where it lies -- no copy. */
int kc_shard_commit(kc_ctx *ctx, const uint64_t *d_segment, uint64_t nwords);
/* Distinct k-mers this shard's regions hold in this flow before they start spilling to the global table: the shard
 * builds regions only for the buckets it owns, 1/rank_n of the geometry's (at most 2^20 regions of at most 4096 LDS slots
 * for all shards together).  A caller whose shards expect more than this each should stay with kc_extract_partition +
 * kc_insert_records, where every shard uses the whole geometry. */
int kc_shard_capacity(kc_ctx *ctx, uint64_t *max_distinct);
/* The shard that owns a canonical k-mer in this flow (role of KmerDHT::get_kmer_target_rank, kmer_dht.cpp:192-196).
 * Depends on the context's geometry: ask a context of the exchange, not kc_owner. */
int kc_shard_owner(kc_ctx *ctx, const uint64_t *kmer_words, int *owner);

/* ---- the reference's wire format (runs mixed with unmodified MHM2 ranks) -------------------------------- */
/* kcount_gpu::SupermerInfo (src/kcount/kcount-gpu/parse_and_pack.hpp:50-54), same layout */
typedef struct kc_supermer {
  int32_t target; /* KmerDHT::get_kmer_target_rank of the supermer's k-mers */
  int32_t offset; /* first character in the block: the left neighbour of its first k-mer */
  uint16_t len;   /* characters: its k-mers + k + 1 */
} kc_supermer;
/*
 * ParseAndPackGPUDriver::process_seq_block + pack_seq_block (parse_and_pack.cpp:281-336): the supermers of a '_'-joined,
 * case-masked block with the CPU backend's semantics (SeqBlockInserter::process_seq, kcount_cpu.cpp:73-103: maximal
 * runs of k-mers with both neighbours and one target rank) and the block packed two characters per byte with the
 * reference's nibble codes (parse_and_pack.cpp:196-213).  out: capacity entries (host), *n_out = how many there are
 * (KC_ERR_CAPACITY if more than capacity: nothing else is lost, call again with room); *num_valid_kmers = k-mers
 * covered; packed_out: (len + 1) / 2 bytes (host) or NULL.  The context's rank_n gives the number of targets.
 */
int kc_build_supermers(kc_ctx *ctx, const char *seqs, uint64_t len, int on_device, kc_supermer *out, uint32_t capacity,
                       uint32_t *n_out, uint32_t *num_valid_kmers, uint8_t *packed_out);
/*
 * HashTableGPUDriver::insert_supermer_block (gpu_hash_table.cpp:655-679): 4-bit packed supermers as
 * src/kcount/kcount_gpu.cpp:153-161 cuts them (odd nibbles masked to 0), joined by the byte '_' as
 * HashTableGPUDriver::insert_supermer joins them; unpacked on the device (gpu_unpack_supermer_block's role) and
 * inserted like kc_submit_seq_block.
 */
int kc_submit_packed_supermers(kc_ctx *ctx, const uint8_t *packed, uint64_t len, int on_device);

/* The contig k-mer pass (dead in the proxy; the backend surface is complete with it):
 * HashTableInserter::init_ctg_kmers (kmer_dht.hpp:103, kcount_cpu.cpp:472-475; HashTableGPUDriver::init_ctg_kmers,
 * gpu_hash_table.hpp:158) -- room for max_ctg_kmers distinct contig k-mers.  Call before kc_finalize. */
int kc_begin_ctg_kmers(kc_ctx *ctx, uint64_t max_ctg_kmers);
/* SeqBlockInserter::process_seq(ctg->seq, depth) + insert_supermer in the contig pass (kcount.cpp:129,
 * kcount_cpu.cpp:357-407; insert_supermer_block with depths, gpu_hash_table.cpp:655-695): a '_'-joined block of contigs
 * and, per character, the depth of its contig (the layout of SeqBlockInserterState::depth_block, kcount_gpu.cpp:74-91).
 * kc_finalize then returns what the reference's insert_into_local_hashtable would after inserting the contigs behind
 * the reads: the reads' results, plus the contig k-mers that are not among them, whose occurrences agree on both
 * extensions (both bases) and all have a depth of 2 or more, with the smallest depth as their count (kc_ctg.hpp).
 * KC_ERR_BAD_BASE for a character outside ACGTN anywhere in the block (the reference DIEs), KC_ERR_CAPACITY when more
 * distinct k-mers came than kc_begin_ctg_kmers made room for (the table is never filled beyond three quarters: a block is
 * taken in as many launches as its free room asks for).  A context that is one of several ranks (rank_n > 1) keeps of a
 * block only the k-mers the read path would keep there -- its share by the k-mer hash, by the reference's target rank
 * (KC_FLAG_REFERENCE_OWNER) or, in the shard flow, by the owner of the k-mer's level-1 bucket -- so every rank may be
 * given every contig, or (like the C++ driver, whose host routes supermers by target) only its own. */
int kc_submit_ctg_block(kc_ctx *ctx, const char *seqs, const uint16_t *depths, uint64_t len, int on_device);
/* The contig pass so far: distinct contig k-mers in its table (what done_ctg_kmer_inserts reports as new inserts,
 * gpu_hash_table.cpp:697-734) and characters submitted.  Either pointer may be null. */
int kc_ctg_stats(kc_ctx *ctx, uint64_t *distinct, uint64_t *positions);

/* KmerDHT::flush_updates -> HashTableInserter::flush_inserts (kmer_dht.cpp:252-258): wait for submitted work. */
int kc_flush(kc_ctx *ctx);

/* HashTableGPUDriver::done_all_inserts + the S7/S8 pass of
 * HashTableInserter::insert_into_local_hashtable (gpu_hash_table.cpp:736-784,
 * kcount_cpu.cpp:523-601): vote, purge, compact.  out may be NULL. */
int kc_finalize(kc_ctx *ctx, kc_result *out);
/* begin_iterate/get_next_entry in bulk: copy the results to host arrays sized from kc_result.n. */
int kc_copy_results(kc_ctx *ctx, uint64_t *keys, uint16_t *counts, uint8_t *left, uint8_t *right);
/* The same results in the shape HashTableGPUDriver hands them to its host (output_keys / output_vals,
 * gpu_hash_table.cpp:776-784, gpu_hash_table.hpp:64-75): keys[n*num_longs] and one 8-byte kc_count_exts per entry
 * (= kcount_gpu::CountExts: uint32 count, int8 left, int8 right), packed on the device and copied once, so that a
 * driver's get_next_entry can point straight into the two arrays.  Either pointer may be NULL. */
typedef struct kc_count_exts {
  uint32_t count;
  int8_t left, right;
  int8_t pad[2];
} kc_count_exts;
int kc_copy_results_entries(kc_ctx *ctx, uint64_t *keys, kc_count_exts *vals);
/* KmerDHT::kmer_exists / get_kmer_count / get_local_kmer_counts (src/kcount/kmer_dht.cpp:198-245) in bulk, against the
 * results kept in HBM: nq k-mers of num_longs words each, in either orientation; counts[i] = 0 (and left/right = 0)
 * when the k-mer did not survive.  The index over the results is built on the first call after kc_finalize.
 * left/right may be NULL.  on_device != 0: all pointers are device pointers. */
int kc_lookup(kc_ctx *ctx, const uint64_t *queries, uint64_t nq, int on_device, uint16_t *counts, uint8_t *left, uint8_t *right);
/* Every table entry before the purge, for tests of S5/S6: keys[n*num_longs], counts[n] (clipped to 65535),
 * exts[n*8] = left ACGT then right ACGT.  Call with NULLs to get n. */
int kc_dump_table(kc_ctx *ctx, uint64_t *keys, uint16_t *counts, uint16_t *exts, uint64_t *n);

int kc_get_stats(kc_ctx *ctx, kc_stats *out);

/* Geometry of the bucketed insert path; 0 in a field keeps the automatic choice.  Only for tests
 * (forcing the overflow paths with tiny capacities) and tuning runs; call right after
 * kc_create / kc_reset, before the first submit. */
typedef struct kc_tuning {
  uint32_t mode;          /* 0 auto (bucketed; compact records where k and the geometry allow), 1 global-table path only,
                             2 bucketed with wide records only */
  uint32_t writers;       /* level-1 writer workgroups (<= 512) */
  uint32_t p1, p2;        /* fan-out of level 1 / level 2: 1..1024 each */
  uint32_t slots;         /* LDS slots per region */
  uint32_t chunk1, chunk2;         /* records per chunk of the level-1 / level-2 chains: powers of two */
  uint32_t chain1_max, chain2_max; /* longest chain, in chunks, of a (writer,bucket) segment / a region */
  uint32_t arena1;        /* chunks in each writer's arena */
  uint64_t ovf_capacity;  /* records per overflow list */
} kc_tuning;
int kc_set_tuning(kc_ctx *ctx, const kc_tuning *t);

/* Per-kernel device time, measured with HIP events recorded on the stream the kernels are
 * launched on (role of the reference's GPUTimer / get_elapsed_time, gpu_common.hpp:83-105,
 * gpu_hash_table.hpp:172).  Needs KC_FLAG_TIME_KERNELS.  Fills up to max entries, *n = how many. */
typedef struct kc_kernel_time {
  char name[48];
  uint64_t launches;
  double total_ms;
} kc_kernel_time;
int kc_get_kernel_times(kc_ctx *ctx, kc_kernel_time *out, int max, int *n);
/* TB/s at which the level-1 arena this context chose took level 1's write pattern for a millisecond when it was
 * allocated (the better of two allocations is kept: which physical memory the driver hands out decides the rate, see
 * DESIGN.md section 5); 0 when no probe ran (arenas under a GiB, KC_ARENA_PROBE=0, the global-table path).  Lets a
 * caller see a slow draw; the library asks no absolute rate of an arena. */
int kc_arena_probe_rate(kc_ctx *ctx, double *tbps);
int kc_clear_kernel_times(kc_ctx *ctx);

/* ---- synthetic ArcticSynth-shaped reads (bench / tests) ------- */
typedef struct kc_synth_params {
  uint64_t seed;
  uint32_t num_genomes;    /* default 64 */
  uint32_t read_len;       /* default 150 */
  uint64_t min_genome_len; /* default 2,000,000 */
  uint64_t max_genome_len; /* default 6,000,000 */
  double sub_error_rate;   /* default 0.005 */
  double lowq_rate;        /* extra low-quality bases, default 0.01 */
  double n_rate;           /* default 0 */
  double abundance_sigma;  /* log-normal sigma, default 1.0 */
} kc_synth_params;

void kc_synth_default_params(kc_synth_params *p);
/* Reads [first_read, first_read+nreads) of the stream defined by p, written as
 * fixed-length records: bases/quals get nreads*read_len bytes, offsets nreads+1
 * entries (relative to this block).  The host and device versions produce the
 * same bytes. */
int kc_synth_reads_host(const kc_synth_params *p, uint64_t first_read, uint64_t nreads, uint8_t *bases, uint8_t *quals,
                        uint64_t *offsets);
int kc_synth_reads_device(kc_ctx *ctx, const kc_synth_params *p, uint64_t first_read, uint64_t nreads, uint8_t *d_bases,
                          uint8_t *d_quals, uint64_t *d_offsets);

#ifdef __cplusplus
}
#endif
#endif /* KCOUNT_MI355_H */
