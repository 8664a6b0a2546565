// kc_common.hpp -- arithmetic shared by host and device code of libkcount_mi355:
// 2-bit k-mer words, reverse complement, the shard/slot hash, the synthetic read
// stream.  Compiled by hipcc for both sides (KC_HD).
//
// Layout of a k-mer (same as the reference's Kmer<MAX_K>, src/kmer.cpp:228-236,255):
// base j sits in bits [63-2(j%32)-1 .. 63-2(j%32)] of word j/32, A0 C1 G2 T3,
// unused trailing bits zero, num_longs = k/32+1 words (src/main.cpp:169-190).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KC_HD __host__ __device__ __forceinline__
#else
#define KC_HD static inline
#endif

#define KC_MAX_LONGS 4
#define KC_QUAL_CUTOFF 20   // KCOUNT_QUAL_CUTOFF, CMakeDefinitions.txt:58
#define KC_EXT_NONE 4u      // extension code for '0' / 'N' (ignored by ExtCounts::inc, kcount_cpu.cpp:157-164)
#define KC_EXT_BITS 6
#define KC_EXT_MASK 0x3FULL
#define KC_COUNT_MAX 65535u // kmer_count_t = uint16_t, kmer_dht.hpp:54

// S3: A0 C1 G2 T3, N (0x4E) -> 2, case-insensitive.  Same truth table as
// x=(c&4)>>1; x+((x^(c&2))>>1) of src/kmer.cpp:191-192.
KC_HD uint32_t kc_base_code(uint32_t c) {
  uint32_t b = (c >> 1) & 3u;
  return b ^ (b >> 1);
}

KC_HD bool kc_is_acgt(uint32_t c) {
  uint32_t u = c & 0xDFu;  // upper-case
  return u == 'A' || u == 'C' || u == 'G' || u == 'T';
}

KC_HD bool kc_is_base_char(uint32_t c) {
  uint32_t u = c & 0xDFu;
  return u == 'A' || u == 'C' || u == 'G' || u == 'T' || u == 'N';
}

// reverse the 32 2-bit groups of a word and complement them
KC_HD uint64_t kc_rc_word(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  v = __brevll(v);
  v = ((v >> 1) & 0x5555555555555555ULL) | ((v & 0x5555555555555555ULL) << 1);
#else
  v = ((v >> 2) & 0x3333333333333333ULL) | ((v & 0x3333333333333333ULL) << 2);
  v = ((v >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((v & 0x0F0F0F0F0F0F0F0FULL) << 4);
  v = __builtin_bswap64(v);
#endif
  return ~v;
}

// mask keeping the bases of word j of a k-mer
KC_HD uint64_t kc_word_mask(int k, int j) {
  int nb = k - 32 * j;
  if (nb >= 32) return ~0ULL;
  if (nb <= 0) return 0ULL;
  return ~(~0ULL >> (2 * nb));
}

// S4 helper: reverse complement of a packed k-mer (src/kmer.cpp:490-510)
template <int NL>
KC_HD void kc_revcomp(const uint64_t (&w)[NL], int k, uint64_t (&out)[NL]) {
  const int ll = (k + 31) / 32;  // words that hold bases
  // t[i] = rc(w[ll-1-i]) for i < ll; written with compile-time indices only (a run-time index would
  // push the arrays to scratch memory)
  uint64_t t[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) {
    uint64_t v = 0;
#pragma unroll
    for (int j = 0; j < NL; j++)
      if (j == ll - 1 - i) v = kc_rc_word(w[j]);
    t[i] = v;
  }
  const int sh = 64 * ll - 2 * k;  // 0..62
  if (sh) {
#pragma unroll
    for (int j = 0; j < NL; j++) {
      uint64_t nxt = (j + 1 < NL) ? t[j + 1] : 0ULL;
      t[j] = (t[j] << sh) | (nxt >> (64 - sh));
    }
  }
#pragma unroll
  for (int j = 0; j < NL; j++) out[j] = (j < ll) ? t[j] : 0ULL;
}

template <int NL>
KC_HD bool kc_less(const uint64_t (&a)[NL], const uint64_t (&b)[NL]) {
#pragma unroll
  for (int j = 0; j < NL; j++) {
    if (a[j] < b[j]) return true;
    if (a[j] > b[j]) return false;
  }
  return false;
}

// One multiply-and-fold step on a pair of 32-bit words (the "mum" of wyhash32): the 64-bit product of the two
// words, each whitened by a constant, split back into its halves.  A 32x32->64 multiply is one quarter-rate VALU
// instruction on gfx950; a 64x64 multiply costs four of them, and the hash runs once per k-mer occurrence in each
// of the three bucketed kernels.
KC_HD void kc_mum32(uint32_t &a, uint32_t &b, uint32_t ca, uint32_t cb) {
  const uint64_t c = (uint64_t)(a ^ ca) * (uint64_t)(b ^ cb);
  a = (uint32_t)c;
  b = (uint32_t)(c >> 32);
}

// 64-bit hash of a canonical k-mer (extension bits already cleared).  The words of the key are absorbed one mum
// each; two further, independent mums of the state give the two halves of the hash, each the xor of a product's
// halves (the high half of a product alone is far from uniform).  The high 32 bits pick the owner shard and the LDS
// slot, the low 32 bits the two bucket levels.  Any deterministic function gives the same final set (F3).
template <int NL>
KC_HD uint64_t kc_hash(const uint64_t (&key)[NL]) {
  uint32_t a = 0x9E3779B9u, b = 0x85EBCA6Bu;
#pragma unroll
  for (int j = 0; j < NL; j++) {
    a ^= (uint32_t)key[j];
    b ^= (uint32_t)(key[j] >> 32);
    kc_mum32(a, b, 0x53c5ca59u, 0x74743c1bu);
  }
  uint32_t a2 = a, b2 = b;
  kc_mum32(a, b, 0x53c5ca59u, 0x74743c1bu);
  kc_mum32(a2, b2, 0xa0761d65u, 0xe7037ed1u);
  return ((uint64_t)(a2 ^ b2) << 32) | (uint64_t)(a ^ b);
}

// ---- an invertible mix of a short k-mer (k <= 23): three Feistel rounds on its two k-bit halves -------------------
// The bucketed path of one-word k-mers splits and probes on the bits of this value instead of a hash of the k-mer:
// because the map is a bijection, the bits that name a record's bucket and region need not be stored with it, which
// halves the records of the second level, and the k-mer is recovered from (region, remaining bits) when the results
// are written.  Round function: the top k bits of the low 32 bits of x * C + D (24-bit multiply: full VALU rate).
// Three rounds (L, R, L): bucket and region are bits of L, which a fourth round (of R) does not touch, and the slot bits
// -- the low bits of R after one round keyed by a twice-mixed L -- spread like random ones over low-complexity families
// too (tests/cpp/test_mix.cpp: poly-A neighbourhoods, tandem repeats, two-letter sequences, families that differ in
// five bases at either end); the fourth round cost level 1 three vector instructions per k-mer, half a millisecond.
KC_HD uint32_t kc_feistel_f(uint32_t x, int i, int k) {
  const uint32_t C[3] = {0x9E3779u, 0x85EBCBu, 0xC2B2AFu}, D[3] = {0x7F4A7Cu, 0x165667u, 0x3C6EF3u};
#if defined(__HIP_DEVICE_COMPILE__)
  return (__umul24(x, C[i]) + D[i]) >> (32 - k);
#else
  return (x * C[i] + D[i]) >> (32 - k);
#endif
}
KC_HD uint64_t kc_feistel_fwd(uint64_t v, int k) {  // v < 4^k
  uint32_t L = (uint32_t)(v >> k), R = (uint32_t)v & ((1u << k) - 1u);
  L ^= kc_feistel_f(R, 0, k);
  R ^= kc_feistel_f(L, 1, k);
  L ^= kc_feistel_f(R, 2, k);
  return ((uint64_t)L << k) | R;
}
KC_HD uint64_t kc_feistel_inv(uint64_t m, int k) {
  uint32_t L = (uint32_t)(m >> k), R = (uint32_t)m & ((1u << k) - 1u);
  L ^= kc_feistel_f(R, 2, k);
  R ^= kc_feistel_f(L, 1, k);
  L ^= kc_feistel_f(R, 0, k);
  return ((uint64_t)L << k) | R;
}
constexpr int KC_COMPACT_MAX_K = 23;

KC_HD uint32_t kc_owner_of_hash(uint64_t h, uint32_t rank_n) {
  return (uint32_t)(((h >> 32) * (uint64_t)rank_n) >> 32);
}

// ---- the reference's own partition function (for runs mixed with unmodified MHM2 ranks) -----------------------
// target rank = quick_hash(minimizer(kmer, m)) % rank_n, the minimizer being the GREATEST over the k-m+1 positions of
// the LEAST of the forward m-mer and the reverse-complement m-mer at the mirrored position
// (src/kmer.cpp:349-398,459-468; src/hash_funcs.c:332-342; src/kcount/kmer_dht.cpp:117-119,192-196).
KC_HD uint64_t kc_quick_hash(uint64_t v) {
  v = v * 3935559000370003845ULL + 2691343689449507681ULL;
  v ^= v >> 21;
  v ^= v << 37;
  v ^= v >> 4;
  v *= 4768777513237032717ULL;
  v ^= v << 20;
  v ^= v >> 41;
  v ^= v << 5;
  return v;
}

KC_HD int kc_minimizer_len(int k) {
  int m = k * 2 / 3 + 1;
  if (m < 15) m = 15;
  if (m > 27) m = 27;
  return m < k ? m : k;
}

// the m-mer starting at base i of a packed k-mer, MSB-aligned and masked to m bases (word picked by compile-time
// indices only: a run-time array index would go to scratch memory on the device)
template <int NL>
KC_HD uint64_t kc_mmer_at(const uint64_t (&w)[NL], int i, int m) {
  const int l = i >> 5, sh = 2 * (i & 31);
  uint64_t hi = 0, lo = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) {
    if (j == l) hi = w[j];
    if (j == l + 1) lo = w[j];
  }
  const uint64_t t = sh ? ((hi << sh) | (lo >> (64 - sh))) : hi;
  return t & (~0ULL << (64 - 2 * m));
}

// f = a k-mer, r = its reverse complement (either order: the minimizer is the same for both strands)
template <int NL>
KC_HD uint32_t kc_reference_owner(const uint64_t (&f)[NL], const uint64_t (&r)[NL], int k, uint32_t rank_n) {
  const int m = kc_minimizer_len(k), ncand = k - m + 1;
  uint64_t best = 0;
  for (int i = 0; i < ncand; i++) {
    const uint64_t a = kc_mmer_at<NL>(f, i, m), b = kc_mmer_at<NL>(r, ncand - 1 - i, m);
    const uint64_t least = a < b ? a : b;
    if (least > best) best = least;
  }
  return (uint32_t)(kc_quick_hash(best) % (uint64_t)rank_n);
}

// ---- synthetic read stream ---------------------------------
#define KC_SYNTH_MAX_GENOMES 1024

struct kc_synth_table {
  uint64_t seed;
  uint32_t num_genomes;
  uint32_t read_len;
  uint64_t err_thr;   // thresholds on a 64-bit uniform
  uint64_t lowq_thr;  // err + lowq
  uint64_t n_thr;
  uint64_t genome_len[KC_SYNTH_MAX_GENOMES];
  uint64_t genome_seed[KC_SYNTH_MAX_GENOMES];
  uint64_t cum[KC_SYNTH_MAX_GENOMES];  // cumulative sampling weight scaled to 2^64-1
};

KC_HD uint64_t kc_splitmix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}

KC_HD uint32_t kc_genome_base(uint64_t gseed, uint64_t pos) { return (uint32_t)(kc_splitmix(gseed + pos * 0xD1342543DE82EF95ULL) >> 62); }

// one base + quality of read r at offset i
KC_HD void kc_synth_base(const kc_synth_table *t, uint64_t rstate, uint32_t g, uint64_t start, bool rev, uint32_t i,
                         uint8_t *base, uint8_t *qual) {
  const uint32_t L = t->read_len;
  uint64_t pos = rev ? start + (L - 1 - i) : start + i;
  uint32_t b = kc_genome_base(t->genome_seed[g], pos);
  if (rev) b = 3u - b;
  uint64_t u = kc_splitmix(rstate + 0x632BE59BD9B4E019ULL * (uint64_t)(i + 1));
  uint8_t q = 'I';
  if (u < t->err_thr) {
    b = (b + 1u + (uint32_t)((u >> 7) % 3u)) & 3u;
    q = '#';
  } else if (u < t->lowq_thr) {
    q = '#';
  }
  uint8_t c = (uint8_t)("ACGT"[b]);
  if (t->n_thr) {
    uint64_t u2 = kc_splitmix(u ^ 0xA5A5A5A5A5A5A5A5ULL);
    if (u2 < t->n_thr) {
      c = 'N';
      q = '#';
    }
  }
  *base = c;
  *qual = q;
}

KC_HD void kc_synth_read_header(const kc_synth_table *t, uint64_t r, uint64_t *rstate, uint32_t *g, uint64_t *start,
                                bool *rev) {
  uint64_t s = kc_splitmix(t->seed ^ (r * 0x9E3779B97F4A7C15ULL));
  uint64_t u1 = kc_splitmix(s + 1), u2 = kc_splitmix(s + 2), u3 = kc_splitmix(s + 3);
  uint32_t lo = 0, hi = t->num_genomes - 1;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (u1 <= t->cum[mid]) hi = mid; else lo = mid + 1;
  }
  *g = lo;
  uint64_t span = t->genome_len[lo] - t->read_len + 1;
  *start = u2 % span;
  *rev = (u3 >> 63) != 0;
  *rstate = s;
}
