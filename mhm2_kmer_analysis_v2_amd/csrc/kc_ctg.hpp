// kc_ctg.hpp -- the contig k-mer pass (HashTableInserter::init_ctg_kmers + insert_supermer in the contig pass,
// src/kcount/kcount_cpu.cpp:357-407,472-475; GPU twin gpu_hash_table.cpp:158-203,697-734).  Dead in the proxy (SURVEY.md
// F8: add_ctg_kmers is commented out, kcount.cpp:106-139) and off the metric: built for completeness of the backend
// surface, exact, not tuned.
//
// The reference inserts the contigs' k-mers AFTER every read, one at a time, into the table the reads built
// (insert_supermer_from_ctg).  Read statement by statement its outcome does not depend on the order of the contig k-mers:
//   * a k-mer the reads keep -- count >= 2 and both extensions voted (neither X nor F) -- ignores every contig k-mer;
//   * any other read entry (a singleton, or a fork / no-vote on either side) is replaced by the first contig k-mer;
//   * contig occurrences among themselves: the entry holds {count, left, right} of the last one taken; a later
//     occurrence with other extensions sets the count to 0 for good, one with the same extensions sets it to the smaller
//     of the two counts -- where "the same" compares against get_ext of an entry whose only counter is its count, which is
//     the base itself only from a count of 2 upwards (dmin_thres), so one occurrence with a count below 2 also ends at
//     0 or 1; what insert_into_local_hashtable finally keeps is count >= 2 with both extensions a base.
// Hence: result = the reads' results, plus every k-mer that is not among them, whose contig occurrences all carry the same
// pair of extensions, both of them bases, and all have a count of 2 or more -- with the smallest of those counts.
// (tests/test_gpu_ctg.py checks this against the oracle's statement-for-statement restatement, contigs in random order.)
//
// Device side: the occurrences go into a table of their own -- per k-mer the smallest count (an atomic max of its
// complement) and the extension pair (first writer sets it, anybody who differs marks the conflict) -- and kc_finalize
// appends the k-mers that qualify and are not among the reads' results (the lookup index of kc_lookup finds those).
#pragma once
#include "kc_bucketed.hpp"
#include "kc_supermer.hpp"

namespace kc {

constexpr uint32_t CTG_EXT_CONFLICT = 0xFFFFFFFFu;

// Which contig k-mers a context of several ranks keeps: the ones the read path would keep there (a caller that routes its
// supermers by target, like the C++ driver, submits only those anyway).
enum { CTG_OWN_ALL = 0, CTG_OWN_HASH, CTG_OWN_REFERENCE, CTG_OWN_BUCKET };
struct CtgOwn {
  uint32_t mode, rank_me, rank_n;
  uint32_t own_lo, own_hi;  // CTG_OWN_BUCKET (the shard flow): the level-1 buckets this shard owns
  Geom gm;
};

// One thread per position [p0, p1) of a '_'-joined block of contigs (any case: a lower-case neighbour counts as low
// quality, get_kmers_and_exts kcount_cpu.cpp:308-336); depths[p] = the depth of the contig position p belongs to (the
// layout of SeqBlockInserterState::depth_block, kcount_gpu.cpp:74-91,160).  t.vals: two words per slot.  The host keeps
// the positions of a launch within the table's free room (kc_submit_ctg_block), so the probe loop of table_slot always
// meets an empty slot.
template <int NL>
__global__ void kc_ctg_insert_kernel(const uint8_t *seqs, const uint16_t *depths, uint64_t p0, uint64_t p1, uint64_t len, int k, Table t,
                                     uint64_t *status, CtgOwn own) {
  const uint64_t p = p0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= p1) return;
  // the alphabet, position by position (every thread its own character, whatever becomes of its window): the reference
  // DIEs on a character outside it (kcount_cpu.cpp:481-487)
  if (seqs[p] != '_' && !sm_is_base(seqs[p])) status[0] = 1;
  if (p < 1 || p + k >= len) return;
  for (int i = -1; i <= k; i++)
    if (seqs[p + i] == '_') return;  // the window of the k-mer and its two neighbours must lie inside one contig (S1, S5)
  for (int i = -1; i <= k; i++)
    if (!sm_is_base(seqs[p + i])) return;  // (its own thread reports it)
  constexpr int KL = NL;  // key words (the contig table keeps the k-mer alone: no extension bits in its last word)
  uint64_t f[KL], r[KL];
#pragma unroll
  for (int j = 0; j < KL; j++) f[j] = 0;
  for (int i = 0; i < k; i++) {
    const uint64_t code = kc_base_code(seqs[p + i]);  // N counts as G inside a k-mer (S3)
#pragma unroll
    for (int j = 0; j < KL; j++)
      if (j == (i >> 5)) f[j] |= code << (62 - 2 * (i & 31));
  }
  kc_revcomp<KL>(f, k, r);
  auto ext_code = [](uint8_t c) -> uint32_t {  // 0-3 = ACGT in upper case; 4 = nothing that is counted ('0', N)
    switch (c) {
      case 'A': return 0u;
      case 'C': return 1u;
      case 'G': return 2u;
      case 'T': return 3u;
      default: return 4u;
    }
  };
  uint32_t le = ext_code(seqs[p - 1]), re = ext_code(seqs[p + k]);
  const bool swap = kc_less<KL>(r, f);  // strict: a palindrome keeps the forward extensions (S4)
  if (swap) {
    const uint32_t l2 = re < 4u ? 3u - re : 4u, r2 = le < 4u ? 3u - le : 4u;
    le = l2;
    re = r2;
#pragma unroll
    for (int j = 0; j < KL; j++) f[j] = r[j];
  }
  if (own.mode != CTG_OWN_ALL) {  // f is the canonical k-mer now
    uint32_t o;
    if (own.mode == CTG_OWN_REFERENCE) {
      uint64_t rr[KL];
      kc_revcomp<KL>(f, k, rr);
      o = kc_reference_owner<KL>(f, rr, k, own.rank_n);
    } else if (own.mode == CTG_OWN_HASH) {
      o = kc_owner_of_hash(kc_hash<KL>(f), own.rank_n);
    } else {
      const uint32_t b1 = (KL == 1 && own.gm.cp) ? (uint32_t)(kc_feistel_fwd(f[0] >> (64u - own.gm.k2), k) >> (own.gm.k2 - own.gm.la))
                                                  : hash_b1(kc_hash<KL>(f), own.gm);
      o = (b1 >= own.own_lo && b1 < own.own_hi) ? own.rank_me : own.rank_me + 1u;
    }
    if (o != own.rank_me) return;
  }
  bool is_new;
  const uint64_t slot = table_slot<KL>(t, f, is_new);
  if (is_new) atomicAdd((unsigned long long *)&status[1], 1ULL);
  uint32_t *v = t.vals + slot * 2;
  atomicMax(&v[0], 0xFFFFFFFFu - (uint32_t)depths[p]);  // the smallest count, as the largest complement (zero = no occurrence yet)
  if (le >= 4u || re >= 4u) {
    atomicExch(&v[1], CTG_EXT_CONFLICT);  // an extension that is no base never survives
  } else {
    const uint32_t e = 1u + le + 4u * re;
    const uint32_t old = atomicCAS(&v[1], 0u, e);
    if (old != 0u && old != e) atomicExch(&v[1], CTG_EXT_CONFLICT);
  }
}

// Every contig k-mer that qualifies and is not among the first n_res results is appended behind them.  out_cap = 0:
// only counts (cursor starts at n_res either way).
template <int NL>
__global__ void kc_ctg_merge_kernel(Table t, uint64_t capacity, const uint32_t *index, uint64_t imask, uint64_t n_res, uint64_t *out_keys,
                                    uint16_t *out_counts, uint8_t *out_left, uint8_t *out_right, uint64_t out_cap, int out_nl, uint64_t *cursor,
                                    uint64_t *sum) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= capacity) return;
  uint64_t key[NL];
#pragma unroll
  for (int j = 0; j < NL; j++) key[j] = t.keys[s * NL + j];
  if (key[NL - 1] == KEY_EMPTY) return;
  const uint32_t count = 0xFFFFFFFFu - t.vals[s * 2], e = t.vals[s * 2 + 1];
  if (e == 0u || e == CTG_EXT_CONFLICT || count < 2u) return;
  // among the reads' results?  (their keys have out_nl words: the k-mer's, then zeros)
  uint64_t h = kc_hash<NL>(key) & imask;
  for (;;) {
    const uint32_t x = index[h];
    if (!x) break;
    bool same = true;
#pragma unroll
    for (int j = 0; j < NL; j++) same &= out_keys[(uint64_t)(x - 1) * out_nl + j] == key[j];
    if (same) return;
    h = (h + 1) & imask;
  }
  const uint64_t o = atomicAdd((unsigned long long *)cursor, 1ULL);
  if (o < out_cap) {
#pragma unroll
    for (int j = 0; j < NL; j++) out_keys[o * out_nl + j] = key[j];
    for (int j = NL; j < out_nl; j++) out_keys[o * out_nl + j] = 0;
    out_counts[o] = (uint16_t)min(count, KC_COUNT_MAX);
    out_left[o] = (uint8_t)("ACGT"[(e - 1u) & 3u]);
    out_right[o] = (uint8_t)("ACGT"[(e - 1u) >> 2]);
    atomicAdd((unsigned long long *)sum, (unsigned long long)min(count, KC_COUNT_MAX));
  }
}

}  // namespace kc
