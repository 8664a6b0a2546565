/*
 * kcount_oracle.c -- CPU restatement of the reference *CPU* kcount path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker and the
 * "port" CPU baseline.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product library
 * (mhm2_kmer_analysis_v2_amd/csrc) never links, loads or calls it.
 *
 * What it restates (reference paths are relative to /root/reference):
 *   S1 read admission            src/kcount/kcount.cpp:78, kcount_cpu.cpp:86,98
 *   S2 quality mask              src/kcount/kcount.cpp:80-85 (KCOUNT_QUAL_CUTOFF=20, CMakeDefinitions.txt:58)
 *   S3 2-bit packing             src/kmer.cpp:155-262 (N -> G at :173,191-192)
 *   S4 canonical form            src/kmer.cpp:270-277,490-510; src/kcount/kcount_cpu.cpp:327-333
 *   partition                    src/kmer.cpp:349-398,459-468; src/hash_funcs.c:332-342;
 *                                src/kcount/kmer_dht.cpp:117-119,192-196
 *   supermers (sender)           src/kcount/kcount_cpu.cpp:73-103
 *   S5 occurrences + extensions  src/kcount/kcount_cpu.cpp:308-336, src/utils.cpp:132-159
 *   table insert                 src/kcount/kcount_cpu.cpp:205-268, src/kmer.cpp:470-473,
 *                                src/hash_funcs.c:77-190 (MurmurHash3_x64_128, seed 313)
 *   S6 saturating accumulation   src/kcount/kcount_cpu.cpp:152-164,349-353
 *   S7 extension vote            src/kcount/kcount_cpu.cpp:135-145,173-182
 *   S8 purge, S9 output          src/kcount/kcount_cpu.cpp:523-601; src/kcount/kmer_dht.hpp:62-68
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - MurmurHash3_x64_64 / quick_hash: checked against oracle/_ref (the
 *     reference's own src/hash_funcs.c compiled unmodified) in tests/.
 *   - packing / revcomp / hash / minimizer: checked against the known answers
 *     SURVEY.md section 8c recorded from the reference's src/kmer.cpp.
 *   - S5-S9 end-to-end (src/kcount/kcount_cpu.cpp needs UPC++, unbuildable
 *     here; the reference ships no tests or fixtures for it): PARITY UNPINNED
 *     beyond hand-derived vectors and an independent second restatement
 *     (tests/spec_model.py).
 *
 * Structure mirrors the reference so that it is a fair CPU baseline:
 *   R emulated ranks; reads are parsed by the calling threads, cut into
 *   supermers by minimizer-hash target, shipped as ASCII (case = quality) to
 *   the target rank, which re-derives k-mers and inserts them into a private
 *   prime-capacity linear-probe table with separate key / value arrays.
 *
 * Deliberate differences (none can change results while dropped == 0):
 *   - table capacity is the next prime >= the request, found by trial
 *     division, not looked up in src/kcount/prime.hpp's table;
 *   - a rank's table grows (rehash) when its load passes 0.66, and also at the
 *     point where the reference would drop an insert (MAX_PROBE=100 exhausted),
 *     so that the "no dropped inserts" precondition (kcount_cpu.cpp:266,
 *     507-516) holds for any input;
 *   - k-mers of a read are produced by a rolling 2-bit window, not the
 *     four-phase shift of kmer.cpp:238-260 (same values).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_LONGS 4           /* MAX_BUILD_KMER 128 -> k <= 127 (CMakeLists.txt:259-279) */
#define ORC_QUAL_CUTOFF 20        /* CMakeDefinitions.txt:58 */
#define ORC_HT_MAX_PROBE 100      /* CMakeDefinitions.txt:67 */
#define ORC_COUNT_MAX 65535       /* kmer_count_t = uint16_t, kmer_dht.hpp:54 */

/* ------------------------------------------------------------------ */
/* hashes: src/hash_funcs.c                                            */
/* ------------------------------------------------------------------ */

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

static inline uint64_t fmix64(uint64_t h) {
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdULL;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ULL;
  h ^= h >> 33;
  return h;
}

/* MurmurHash3_x64_128 (public-domain algorithm by A. Appleby), seed 313, low
 * 64 bits: hash_funcs.c:77-170,185-190. */
uint64_t orc_murmur3_x64_64(const void *key, uint32_t len) {
  const uint8_t *data = (const uint8_t *)key;
  const uint32_t nblocks = len / 16;
  uint64_t h1 = 313, h2 = 313;
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  for (uint32_t i = 0; i < nblocks; i++) {
    uint64_t k1, k2;
    memcpy(&k1, data + 16 * i, 8);
    memcpy(&k2, data + 16 * i + 8, 8);
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  const uint8_t *tail = data + nblocks * 16;
  uint64_t k1 = 0, k2 = 0;
  int rem = len & 15;
  for (int i = rem - 1; i >= 8; i--) k2 ^= (uint64_t)tail[i] << (8 * (i - 8));
  if (rem > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
  for (int i = (rem > 8 ? 8 : rem) - 1; i >= 0; i--) k1 ^= (uint64_t)tail[i] << (8 * i);
  if (rem > 0) { k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }
  h1 ^= len; h2 ^= len;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2;
  return h1;
}

/* hash_funcs.c:332-342 */
uint64_t orc_quick_hash(uint64_t v) {
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

/* ------------------------------------------------------------------ */
/* k-mer value type: src/kmer.cpp                                      */
/* ------------------------------------------------------------------ */

/* Kmer<MAX_K> with MAX_K = (k/32+1)*32 (main.cpp:169-190) has (MAX_K+31)/32 words. */
int orc_num_longs(int k) { return k / 32 + 1; }

/* kmer_dht.cpp:117-119 */
int orc_minimizer_len(int k) {
  int m = k * 2 / 3 + 1;
  if (m < 15) m = 15;
  if (m > 27) m = 27;
  return m;
}

/* S3: kmer.cpp:191-192.  A0 C1 G2 T3; 'N' (and its lowercase) lands on 2. */
static inline uint64_t base_code(char c) {
  uint64_t x = ((uint64_t)(c & 4)) >> 1;
  return x + ((x ^ (uint64_t)(c & 2)) >> 1);
}

/* pack the k characters at s into nl MSB-first words, unused bits zero. */
void orc_pack_kmer(const char *s, int k, uint64_t *out) {
  int nl = orc_num_longs(k);
  for (int l = 0; l < nl; l++) out[l] = 0;
  for (int i = 0; i < k; i++) out[i / 32] |= base_code(s[i]) << (2 * (31 - (i % 32)));
}

/* all k-mers of a sequence (kmer.cpp:169-262); returns how many. */
int orc_get_kmers(const char *seq, int len, int k, uint64_t *out) {
  if (len < k) return 0;
  int nl = orc_num_longs(k);
  int n = len - k + 1;
  uint64_t w[ORC_MAX_LONGS];
  orc_pack_kmer(seq, k, w);
  memcpy(out, w, 8 * nl);
  int lw = (k - 1) / 32;               /* word holding the last base */
  int lsh = 2 * (31 - ((k - 1) % 32)); /* its bit position */
  for (int i = 1; i < n; i++) {
    for (int l = 0; l < nl; l++) {
      uint64_t carry = (l + 1 < nl) ? (w[l + 1] >> 62) : 0;
      w[l] = (w[l] << 2) | carry;
    }
    /* the shift moved the old last base one slot left; clear what now sits past
     * the end, then drop in the new base */
    w[lw] &= ~(3ULL << lsh);
    w[lw] |= base_code(seq[i + k - 1]) << lsh;
    for (int l = lw + 1; l < nl; l++) w[l] = 0;
    memcpy(out + (size_t)i * nl, w, 8 * nl);
  }
  return n;
}

/* reverse the 32 2-bit groups of a word and complement them */
static inline uint64_t rc_word(uint64_t v) {
  v = ((v >> 2) & 0x3333333333333333ULL) | ((v & 0x3333333333333333ULL) << 2);
  v = ((v >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((v & 0x0F0F0F0F0F0F0F0FULL) << 4);
  v = __builtin_bswap64(v);
  return ~v;
}

/* kmer.cpp:490-510 */
void orc_revcomp(const uint64_t *in, int k, uint64_t *out) {
  int nl = orc_num_longs(k);
  int last_long = (k + 31) / 32;
  uint64_t t[ORC_MAX_LONGS] = {0, 0, 0, 0};
  for (int i = 0; i < last_long; i++) t[last_long - 1 - i] = rc_word(in[i]);
  int shift = (k % 32) ? 2 * (32 - (k % 32)) : 0;
  if (shift) {
    for (int i = 0; i < last_long; i++) {
      uint64_t nxt = (i + 1 < last_long) ? t[i + 1] : 0;
      t[i] = (t[i] << shift) | (nxt >> (64 - shift));
    }
  }
  for (int l = 0; l < nl; l++) out[l] = (l < last_long) ? t[l] : 0;
}

/* kmer.cpp:270-277 */
int orc_kmer_less(const uint64_t *a, const uint64_t *b, int nl) {
  for (int i = 0; i < nl; i++) {
    if (a[i] < b[i]) return 1;
    if (a[i] > b[i]) return 0;
  }
  return 0;
}

static inline int kmer_eq(const uint64_t *a, const uint64_t *b, int nl) {
  for (int i = 0; i < nl; i++)
    if (a[i] != b[i]) return 0;
  return 1;
}

/* kmer.cpp:470-473 */
uint64_t orc_kmer_hash(const uint64_t *kmer, int nl) { return orc_murmur3_x64_64(kmer, (uint32_t)(nl * 8)); }

/* the m-mer starting at base i of a packed k-mer, MSB-aligned, masked to m bases */
static inline uint64_t mmer_at(const uint64_t *w, int nl, int i, int m) {
  int l = i / 32, sh = 2 * (i % 32);
  uint64_t t = w[l] << sh;
  if (sh && l + 1 < nl) t |= w[l + 1] >> (64 - sh);
  return t & (~0ULL << (64 - 2 * m));
}

/* kmer.cpp:349-398 with revcomp given: max over positions of min(fwd m-mer,
 * rc m-mer at the mirrored position). */
uint64_t orc_minimizer(const uint64_t *kmer, int k, int m) {
  int nl = orc_num_longs(k);
  uint64_t rc[ORC_MAX_LONGS];
  orc_revcomp(kmer, k, rc);
  int ncand = k - m + 1;
  uint64_t best = 0;
  for (int i = 0; i < ncand; i++) {
    uint64_t f = mmer_at(kmer, nl, i, m);
    uint64_t r = mmer_at(rc, nl, ncand - 1 - i, m);
    uint64_t least = f < r ? f : r;
    if (least > best) best = least;
  }
  return best;
}

/* kmer.cpp:459-468 + kmer_dht.cpp:192-196 */
uint64_t orc_minimizer_hash(const uint64_t *kmer, int k, int m) { return orc_quick_hash(orc_minimizer(kmer, k, m)); }

int orc_target_rank(const uint64_t *kmer, int k, int m, int nranks) {
  return (int)(orc_minimizer_hash(kmer, k, m) % (uint64_t)nranks);
}

/* utils.cpp:132-159 restricted to what can reach it on this path */
static inline char comp_nucleotide(char c) {
  switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    case 'N': return 'N';
    default: return '0';
  }
}

/* ------------------------------------------------------------------ */
/* S7: kcount_cpu.cpp:135-145,173-182                                  */
/* ------------------------------------------------------------------ */
char orc_get_ext(const uint16_t c[4], uint16_t count, int dmin_thres) {
  static const char L[4] = {'A', 'C', 'G', 'T'};
  int idx[4] = {0, 1, 2, 3};
  /* sort by count desc, ties by letter desc */
  for (int i = 0; i < 4; i++)
    for (int j = i + 1; j < 4; j++) {
      int a = idx[i], b = idx[j];
      int swap = (c[a] == c[b]) ? (L[a] < L[b]) : (c[a] < c[b]);
      if (swap) { idx[i] = b; idx[j] = a; }
    }
  int top = c[idx[0]], runner = c[idx[1]];
  /* DYN_MIN_DEPTH = 0.9 (CMakeDefinitions.txt:70): double arithmetic, truncated */
  int dmin_dyn = (int)((1.0 - 0.9) * count);
  if (dmin_dyn < dmin_thres) dmin_dyn = dmin_thres;
  if (top < dmin_dyn) return 'X';
  if (runner >= dmin_dyn) return 'F';
  return L[idx[0]];
}

/* ------------------------------------------------------------------ */
/* per-rank table: kcount_cpu.cpp:191-295                              */
/* ------------------------------------------------------------------ */
typedef struct {
  uint16_t left[4];
  uint16_t right[4];
  uint16_t count;
  uint8_t from_ctg;
} orc_vals; /* 20 bytes like KmerExtsCounts */

typedef struct {
  uint64_t capacity;
  uint64_t num_elems;
  uint64_t num_dropped;
  uint64_t sum_probe, max_probe;
  uint64_t num_grows; /* rehashes so far: a timed run sized by the reference's rule should show 0 */
  uint64_t *keys; /* capacity * nl, 0xFF filled */
  orc_vals *vals;
} orc_table;

static int is_prime(uint64_t n) {
  if (n < 2) return 0;
  if (n % 2 == 0) return n == 2;
  for (uint64_t d = 3; d * d <= n; d += 2)
    if (n % d == 0) return 0;
  return 1;
}

static uint64_t next_prime(uint64_t n) {
  if (n < 3) return 3;
  if (n % 2 == 0) n++;
  while (!is_prime(n)) n += 2;
  return n;
}

static int table_init(orc_table *t, uint64_t min_cap, int nl) {
  memset(t, 0, sizeof(*t));
  t->capacity = next_prime(min_cap < 11 ? 11 : min_cap);
  t->keys = (uint64_t *)malloc(t->capacity * nl * 8);
  t->vals = (orc_vals *)calloc(t->capacity, sizeof(orc_vals));
  if (!t->keys || !t->vals) return -1;
  memset(t->keys, 0xff, t->capacity * nl * 8);
  return 0;
}

static void table_free(orc_table *t) {
  free(t->keys);
  free(t->vals);
  t->keys = NULL;
  t->vals = NULL;
}

/* KmerMapExts::insert without the ctg override pass: kcount_cpu.cpp:232-247,266-267 */
static orc_vals *table_insert(orc_table *t, const uint64_t *kmer, int nl) {
  uint64_t slot = orc_kmer_hash(kmer, nl) % t->capacity;
  uint64_t max_probe = t->capacity < ORC_HT_MAX_PROBE ? t->capacity : ORC_HT_MAX_PROBE;
  for (uint64_t i = 1; i <= max_probe; i++) {
    uint64_t *ks = t->keys + slot * nl;
    if (ks[nl - 1] == ~0ULL) {
      memcpy(ks, kmer, 8 * nl);
      t->sum_probe += i;
      if (i > t->max_probe) t->max_probe = i;
      t->num_elems++;
      return &t->vals[slot];
    } else if (kmer_eq(ks, kmer, nl)) {
      return &t->vals[slot];
    }
    slot = (slot + 1) % t->capacity;
  }
  return NULL; /* no room within MAX_PROBE: the reference counts a drop here (kcount_cpu.cpp:266) */
}

static int table_grow(orc_table *t, int nl) {
  orc_table nt;
  if (table_init(&nt, t->capacity * 2, nl)) return -1;
  for (uint64_t s = 0; s < t->capacity; s++) {
    uint64_t *ks = t->keys + s * nl;
    if (ks[nl - 1] == ~0ULL) continue;
    /* unbounded probe while rehashing: nothing may be lost */
    uint64_t slot = orc_kmer_hash(ks, nl) % nt.capacity;
    while (nt.keys[slot * nl + nl - 1] != ~0ULL) slot = (slot + 1) % nt.capacity;
    memcpy(nt.keys + slot * nl, ks, 8 * nl);
    nt.vals[slot] = t->vals[s];
    nt.num_elems++;
  }
  nt.num_dropped = t->num_dropped;
  nt.num_grows = t->num_grows + 1;
  table_free(t);
  *t = nt;
  return 0;
}

/* ------------------------------------------------------------------ */
/* context                                                             */
/* ------------------------------------------------------------------ */
typedef struct {
  char *buf;
  size_t len, cap;
} bytebuf;

static int bb_push(bytebuf *b, const void *p, size_t n) {
  if (b->len + n > b->cap) {
    size_t nc = b->cap ? b->cap * 2 : 4096;
    while (nc < b->len + n) nc *= 2;
    char *nb = (char *)realloc(b->buf, nc);
    if (!nb) return -1;
    b->buf = nb;
    b->cap = nc;
  }
  memcpy(b->buf + b->len, p, n);
  b->len += n;
  return 0;
}

typedef struct orc_ctx {
  int k, nl, m, qual_offset, dmin_thres, nranks, nthreads;
  orc_table *tables; /* [nranks] */
  bytebuf *outbox;   /* [nthreads * nranks]: supermers in flight (u32 len + chars) */
  /* stats */
  uint64_t num_reads, raw_kmers, num_supermers, supermer_bytes, kmers_inserted;
  /* results, sorted by key */
  uint64_t nres;
  uint64_t *res_keys;
  uint16_t *res_counts;
  char *res_left, *res_right;
  uint64_t num_unique, num_purged, sum_counts, num_dropped;
} orc_ctx;

orc_ctx *orc_create(int k, int qual_offset, int dmin_thres, int nranks, int nthreads, uint64_t capacity_per_rank) {
  if (k < 3 || k >= 32 * ORC_MAX_LONGS || nranks < 1) return NULL;
  orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
  if (!c) return NULL;
  c->k = k;
  c->nl = orc_num_longs(k);
  c->m = orc_minimizer_len(k);
  if (c->m > k) c->m = k;
  c->qual_offset = qual_offset;
  c->dmin_thres = dmin_thres;
  c->nranks = nranks;
#ifdef _OPENMP
  c->nthreads = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
  c->nthreads = 1;
#endif
  c->tables = (orc_table *)calloc(nranks, sizeof(orc_table));
  c->outbox = (bytebuf *)calloc((size_t)c->nthreads * nranks, sizeof(bytebuf));
  if (!c->tables || !c->outbox) return NULL;
  for (int r = 0; r < nranks; r++)
    if (table_init(&c->tables[r], capacity_per_rank ? capacity_per_rank : 1024, c->nl)) return NULL;
  return c;
}

void orc_destroy(orc_ctx *c) {
  if (!c) return;
  for (int r = 0; r < c->nranks; r++) table_free(&c->tables[r]);
  for (int i = 0; i < c->nthreads * c->nranks; i++) free(c->outbox[i].buf);
  free(c->tables);
  free(c->outbox);
  free(c->res_keys);
  free(c->res_counts);
  free(c->res_left);
  free(c->res_right);
  free(c);
}

static int emit_supermer(orc_ctx *c, int tid, int target, const char *s, uint32_t len) {
  bytebuf *b = &c->outbox[(size_t)tid * c->nranks + target];
  if (bb_push(b, &len, 4)) return -1;
  return bb_push(b, s, len);
}

/* SeqBlockInserter::process_seq (kcount_cpu.cpp:73-103) as a list: the supermers of one case-masked read as
 * (target, start, length) over the read.  Returns how many (they never number more than len); kbuf holds the read's
 * k-mers afterwards. */
static int supermers_of(const orc_ctx *c, const char *seq, int len, uint64_t *kbuf, int *targets, int *starts, int *lens) {
  int k = c->k, nl = c->nl, n = 0;
  if (len < k + 2) return 0; /* nothing can be emitted (S1) */
  orc_get_kmers(seq, len, k, kbuf);
  /* targets of the canonical k-mers 1 .. len-k-1 */
  uint64_t rc[ORC_MAX_LONGS];
  int start = 0; /* supermer = seq[start .. end] */
  int prev_target = -1;
  for (int i = 1; i < len - k; i++) {
    uint64_t *km = kbuf + (size_t)i * nl;
    orc_revcomp(km, k, rc);
    const uint64_t *canon = orc_kmer_less(rc, km, nl) ? rc : km;
    int target = orc_target_rank(canon, k, c->m, c->nranks);
    if (i == 1) {
      prev_target = target;
      start = 0;
    } else if (target != prev_target) {
      /* supermer so far covers seq[start .. i+k-1] */
      targets[n] = prev_target; starts[n] = start; lens[n] = i + k - start; n++;
      start = i - 1;
      prev_target = target;
    }
  }
  /* final supermer reaches the end of the read: seq[start .. len-1] */
  targets[n] = prev_target; starts[n] = start; lens[n] = len - start; n++;
  return n;
}

/* the same through the library boundary, for tests of the wire format: arrays of at least len entries */
int orc_build_supermers(const orc_ctx *c, const char *seq, int len, int *targets, int *starts, int *lens) {
  uint64_t *kbuf = (uint64_t *)malloc(((size_t)len + 1) * 8 * c->nl);
  if (!kbuf) return -1;
  int n = supermers_of(c, seq, len, kbuf, targets, starts, lens);
  free(kbuf);
  return n;
}

/* sender: count_kmers body (kcount.cpp:78-87) + SeqBlockInserter::process_seq
 * (kcount_cpu.cpp:73-103).  seq is a private, already case-masked copy. */
static int process_seq(orc_ctx *c, int tid, const char *seq, int len, uint64_t *kbuf, uint64_t *n_super) {
  if (len < c->k + 2) return 0;
  int *tr = (int *)malloc(3 * (size_t)len * sizeof(int));
  if (!tr) return -1;
  int n = supermers_of(c, seq, len, kbuf, tr, tr + len, tr + 2 * len);
  int err = 0;
  for (int i = 0; i < n && !err; i++) {
    if (emit_supermer(c, tid, tr[i], seq + tr[len + i], (uint32_t)tr[2 * len + i])) err = -1;
    (*n_super)++;
  }
  free(tr);
  return err;
}

/* receiver: HashTableInserter::insert_supermer -> insert_supermer_from_read ->
 * get_kmers_and_exts (kcount_cpu.cpp:477-493,338-355,308-336) */
static int insert_supermer(orc_ctx *c, orc_table *t, const char *sm, int len, char *up, uint64_t *kbuf, uint64_t *n_ins) {
  int k = c->k, nl = c->nl;
  for (int i = 0; i < len; i++) {
    char b = sm[i];
    if (b >= 'a' && b <= 'z') b += 'A' - 'a';
    if (b != 'A' && b != 'C' && b != 'G' && b != 'T' && b != 'N') return -2; /* reference DIEs */
    up[i] = b;
  }
  orc_get_kmers(up, len, k, kbuf);
  uint64_t rc[ORC_MAX_LONGS];
  for (int i = 1; i < len - k; i++) {
    const uint64_t *km = kbuf + (size_t)i * nl;
    char left = up[i - 1];
    if (!(sm[i - 1] >= 'A' && sm[i - 1] <= 'Z')) left = '0';
    char right = up[i + k];
    if (!(sm[i + k] >= 'A' && sm[i + k] <= 'Z')) right = '0';
    orc_revcomp(km, k, rc);
    if (orc_kmer_less(rc, km, nl)) {
      km = rc;
      char tl = left;
      left = comp_nucleotide(right);
      right = comp_nucleotide(tl);
    }
    if ((double)(t->num_elems + 1) > 0.66 * (double)t->capacity)
      if (table_grow(t, nl)) return -1;
    orc_vals *v = table_insert(t, km, nl);
    while (!v) { /* where the reference would drop, grow and retry: keeps dropped == 0 */
      if (table_grow(t, nl)) return -1;
      v = table_insert(t, km, nl);
    }
    (*n_ins)++;
    /* S6 */
    int cnt = v->count + 1;
    v->count = (uint16_t)(cnt > ORC_COUNT_MAX ? ORC_COUNT_MAX : cnt);
    int li = left == 'A' ? 0 : left == 'C' ? 1 : left == 'G' ? 2 : left == 'T' ? 3 : -1;
    int ri = right == 'A' ? 0 : right == 'C' ? 1 : right == 'G' ? 2 : right == 'T' ? 3 : -1;
    if (li >= 0 && v->left[li] < ORC_COUNT_MAX) v->left[li]++;
    if (ri >= 0 && v->right[ri] < ORC_COUNT_MAX) v->right[ri]++;
  }
  return 0;
}

/* One block of reads through sender + exchange + receiver.  bases/quals are
 * ASCII, read r is [offsets[r], offsets[r+1]).  Returns 0, -1 (memory), -2 (bad
 * base character). */
int orc_add_reads(orc_ctx *c, const char *bases, const char *quals, const uint64_t *offsets, uint64_t nreads) {
  int err = 0;
  uint64_t raw = 0, nsup = 0, nins = 0;
  size_t maxlen = 0;
  for (uint64_t r = 0; r < nreads; r++) {
    size_t l = offsets[r + 1] - offsets[r];
    if (l > maxlen) maxlen = l;
  }
#pragma omp parallel num_threads(c->nthreads) reduction(+ : raw, nsup, nins)
  {
#ifdef _OPENMP
    int tid = omp_get_thread_num();
#else
    int tid = 0;
#endif
    char *seq = (char *)malloc(maxlen + 1);
    uint64_t *kbuf = (uint64_t *)malloc((maxlen + 1) * 8 * c->nl);
    int lerr = (!seq || !kbuf) ? -1 : 0;
    /* ---- sender side ---- */
#pragma omp for schedule(static)
    for (int64_t r = 0; r < (int64_t)nreads; r++) {
      if (lerr) continue;
      int len = (int)(offsets[r + 1] - offsets[r]);
      if (len < c->k) continue; /* kcount.cpp:78 */
      const char *b = bases + offsets[r], *q = quals + offsets[r];
      for (int i = 0; i < len; i++) {
        char ch = b[i];
        if (ch >= 'a' && ch <= 'z') ch += 'A' - 'a';
        if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T' && ch != 'N') lerr = -2;
        /* quality is clipped to 0..31 by the read cache (packed_reads.cpp:125); 20 < 31 so the test is unchanged */
        if (q[i] < c->qual_offset + ORC_QUAL_CUTOFF) ch += 'a' - 'A'; /* kcount.cpp:81-82 */
        seq[i] = ch;
      }
      if (lerr) continue;
      raw += (uint64_t)(len - c->k + 1); /* kcount.cpp:86 */
      if (process_seq(c, tid, seq, len, kbuf, &nsup)) lerr = -1;
    }
    /* ---- receiver side: rank t drains every sender's outbox for t ---- */
#pragma omp barrier
#pragma omp for schedule(dynamic, 1)
    for (int t = 0; t < c->nranks; t++) {
      if (lerr) continue;
      for (int s = 0; s < c->nthreads; s++) {
        bytebuf *bb = &c->outbox[(size_t)s * c->nranks + t];
        size_t pos = 0;
        while (pos < bb->len) {
          uint32_t sl;
          memcpy(&sl, bb->buf + pos, 4);
          pos += 4;
          int rcode = insert_supermer(c, &c->tables[t], bb->buf + pos, (int)sl, seq, kbuf, &nins);
          if (rcode) lerr = rcode;
          pos += sl;
        }
        bb->len = 0;
      }
    }
    free(seq);
    free(kbuf);
    if (lerr) {
#pragma omp critical
      err = lerr;
    }
  }
  c->num_reads += nreads;
  c->raw_kmers += raw;
  c->num_supermers += nsup;
  c->kmers_inserted += nins;
  return err;
}

/* ------------------------------------------------------------------ */
/* the contig pass (dead in the proxy, SURVEY.md F8: add_ctg_kmers is commented out at kcount.cpp:106-139, but the
 * backend code is there): HashTableInserter::init_ctg_kmers (kcount_cpu.cpp:472-475) switches insert_supermer to
 * insert_supermer_from_ctg (kcount_cpu.cpp:357-407).  A contig's k-mers come with the contig's depth as their count
 * (kcount.cpp:129, Supermer::count, get_kmers_and_exts kcount_cpu.cpp:308-336).                                      */
/* ------------------------------------------------------------------ */
static void vals_set_from_ctg(orc_vals *v, char left, char right, int count) {
  memset(v, 0, sizeof(*v));
  v->count = (uint16_t)count;
  v->from_ctg = 1;
  /* ExtCounts::inc(ext, count), kcount_cpu.cpp:157-164 */
  int li = left == 'A' ? 0 : left == 'C' ? 1 : left == 'G' ? 2 : left == 'T' ? 3 : -1;
  int ri = right == 'A' ? 0 : right == 'C' ? 1 : right == 'G' ? 2 : right == 'T' ? 3 : -1;
  if (li >= 0) v->left[li] = (uint16_t)count;
  if (ri >= 0) v->right[ri] = (uint16_t)count;
}

/* insert_supermer_from_ctg, kcount_cpu.cpp:357-407, statement for statement */
static int insert_supermer_ctg(orc_ctx *c, orc_table *t, const char *sm, int len, int depth, char *up, uint64_t *kbuf) {
  int k = c->k, nl = c->nl;
  for (int i = 0; i < len; i++) {
    char b = sm[i];
    if (b >= 'a' && b <= 'z') b += 'A' - 'a';
    if (b != 'A' && b != 'C' && b != 'G' && b != 'T' && b != 'N') return -2;
    up[i] = b;
  }
  orc_get_kmers(up, len, k, kbuf);
  uint64_t rc[ORC_MAX_LONGS];
  for (int i = 1; i < len - k; i++) {
    const uint64_t *km = kbuf + (size_t)i * nl;
    char left = up[i - 1];
    if (!(sm[i - 1] >= 'A' && sm[i - 1] <= 'Z')) left = '0';
    char right = up[i + k];
    if (!(sm[i + k] >= 'A' && sm[i + k] <= 'Z')) right = '0';
    orc_revcomp(km, k, rc);
    if (orc_kmer_less(rc, km, nl)) {
      km = rc;
      char tl = left;
      left = comp_nucleotide(right);
      right = comp_nucleotide(tl);
    }
    if ((double)(t->num_elems + 1) > 0.66 * (double)t->capacity)
      if (table_grow(t, nl)) return -1;
    uint64_t before = t->num_elems;
    orc_vals *v = table_insert(t, km, nl);
    while (!v) {
      if (table_grow(t, nl)) return -1;
      before = t->num_elems;
      v = table_insert(t, km, nl);
    }
    int is_new = t->num_elems != before;
    int count = depth;
    int insert_it = 0;
    if (is_new) {
      insert_it = 1;
    } else if (!v->from_ctg) {
      /* existing entry is from a read */
      if (v->count == 1) {
        insert_it = 1; /* singleton read k-mer: would be purged anyway */
      } else {
        char le = orc_get_ext(v->left, v->count, c->dmin_thres), re = orc_get_ext(v->right, v->count, c->dmin_thres);
        if (le == 'X' || le == 'F' || re == 'X' || re == 'F') insert_it = 1; /* non-UU: replace */
      }
    } else {
      /* existing entry from a contig */
      if (v->count) {
        insert_it = 1;
        char le = orc_get_ext(v->left, v->count, c->dmin_thres), re = orc_get_ext(v->right, v->count, c->dmin_thres);
        if (le != left || re != right) count = 0;                 /* two contig k-mers disagree: set up to purge */
        else count = count < v->count ? count : v->count;         /* the same k-mer from several contigs: not counted again */
      }
    }
    if (insert_it) vals_set_from_ctg(v, left, right, count);
  }
  return 0;
}

/* One contig through SeqBlockInserter::process_seq(seq, depth) + the exchange + insert_supermer in the contig pass.
 * Call after every read has been added.  Returns 0, -1 (memory), -2 (bad character). */
int orc_add_ctg(orc_ctx *c, const char *seq, int len, int depth) {
  if (len < c->k + 2) return 0; /* kcount.cpp:127 */
  uint64_t *kbuf = (uint64_t *)malloc(((size_t)len + 1) * 8 * c->nl);
  int *tr = (int *)malloc(3 * (size_t)len * sizeof(int));
  char *up = (char *)malloc((size_t)len + 1);
  int err = (!kbuf || !tr || !up) ? -1 : 0;
  if (!err) {
    int n = supermers_of(c, seq, len, kbuf, tr, tr + len, tr + 2 * len);
    for (int i = 0; i < n && !err; i++)
      err = insert_supermer_ctg(c, &c->tables[tr[i]], seq + tr[len + i], tr[2 * len + i], depth, up, kbuf);
  }
  free(kbuf);
  free(tr);
  free(up);
  return err;
}

/* Receiver entry on its own (what the RPC callback calls, kmer_dht.cpp:147-151):
 * one ASCII supermer, case = quality, straight into rank `target`'s table. */
int orc_insert_supermer(orc_ctx *c, int target, const char *sm, int len) {
  if (target < 0 || target >= c->nranks) return -3;
  char *up = (char *)malloc(len + 1);
  uint64_t *kbuf = (uint64_t *)malloc(((size_t)len + 1) * 8 * c->nl);
  uint64_t nins = 0;
  int rc = (!up || !kbuf) ? -1 : insert_supermer(c, &c->tables[target], sm, len, up, kbuf, &nins);
  c->kmers_inserted += nins;
  free(up);
  free(kbuf);
  return rc;
}

typedef struct {
  uint64_t key[ORC_MAX_LONGS];
  uint16_t count;
  char left, right;
} orc_rec;

static int g_sort_nl;
static int rec_cmp(const void *a, const void *b) {
  const orc_rec *x = (const orc_rec *)a, *y = (const orc_rec *)b;
  for (int i = 0; i < g_sort_nl; i++) {
    if (x->key[i] < y->key[i]) return -1;
    if (x->key[i] > y->key[i]) return 1;
  }
  return 0;
}

/* insert_into_local_hashtable (kcount_cpu.cpp:523-601): vote, purge, collect.  Like the reference every rank works
 * on its own table (here: one OpenMP thread per rank at a time) in two scans: count the survivors, then vote, filter and
 * write them to the rank's share of the result arrays (the reference inserts them into its compact local map).  The
 * result arrays are the concatenation over ranks; sort_results != 0 sorts them by key (S9: the tests compare sorted
 * sets; the reference itself never sorts, so the timed baseline does not either). */
int orc_finalize_ex(orc_ctx *c, int sort_results) {
  int nl = c->nl;
  int nr = c->nranks;
  uint64_t *cnt = (uint64_t *)calloc((size_t)nr + 1, 8);
  uint64_t *purged_r = (uint64_t *)calloc((size_t)nr, 8), *sum_r = (uint64_t *)calloc((size_t)nr, 8);
  if (!cnt || !purged_r || !sum_r) return -1;
  /* scan 1: survivors per rank */
#pragma omp parallel for schedule(dynamic, 1) num_threads(c->nthreads)
  for (int r = 0; r < nr; r++) {
    const orc_table *t = &c->tables[r];
    uint64_t n = 0, p = 0;
    for (uint64_t s = 0; s < t->capacity; s++) {
      const uint64_t *ks = t->keys + s * nl;
      if (ks[nl - 1] == ~0ULL) continue;
      const orc_vals *v = &t->vals[s];
      if (v->count < 2) { p++; continue; }
      char l = orc_get_ext(v->left, v->count, c->dmin_thres);
      char rr = orc_get_ext(v->right, v->count, c->dmin_thres);
      if (l == 'X' || l == 'F' || rr == 'X' || rr == 'F') { p++; continue; }
      n++;
    }
    cnt[r + 1] = n;
    purged_r[r] = p;
  }
  uint64_t total = 0, dropped = 0, purged = 0;
  for (int r = 0; r < nr; r++) {
    cnt[r + 1] += cnt[r];
    total += c->tables[r].num_elems;
    dropped += c->tables[r].num_dropped;
    purged += purged_r[r];
  }
  const uint64_t n = cnt[nr];
  free(c->res_keys); free(c->res_counts); free(c->res_left); free(c->res_right);
  c->res_keys = (uint64_t *)malloc((n ? n : 1) * 8 * nl);
  c->res_counts = (uint16_t *)malloc((n ? n : 1) * 2);
  c->res_left = (char *)malloc(n ? n : 1);
  c->res_right = (char *)malloc(n ? n : 1);
  if (!c->res_keys || !c->res_counts || !c->res_left || !c->res_right) return -1;
  /* scan 2: vote, filter, write */
#pragma omp parallel for schedule(dynamic, 1) num_threads(c->nthreads)
  for (int r = 0; r < nr; r++) {
    const orc_table *t = &c->tables[r];
    uint64_t o = cnt[r], sum = 0;
    for (uint64_t s = 0; s < t->capacity; s++) {
      const uint64_t *ks = t->keys + s * nl;
      if (ks[nl - 1] == ~0ULL) continue;
      const orc_vals *v = &t->vals[s];
      if (v->count < 2) continue;
      char l = orc_get_ext(v->left, v->count, c->dmin_thres);
      char rr = orc_get_ext(v->right, v->count, c->dmin_thres);
      if (l == 'X' || l == 'F' || rr == 'X' || rr == 'F') continue;
      memcpy(c->res_keys + o * nl, ks, 8 * nl);
      c->res_counts[o] = v->count;
      c->res_left[o] = l;
      c->res_right[o] = rr;
      sum += v->count;
      o++;
    }
    sum_r[r] = sum;
  }
  uint64_t sum = 0;
  for (int r = 0; r < nr; r++) sum += sum_r[r];
  free(cnt); free(purged_r); free(sum_r);
  if (sort_results && n > 1) {
    orc_rec *recs = (orc_rec *)malloc(n * sizeof(orc_rec));
    if (!recs) return -1;
    for (uint64_t i = 0; i < n; i++) {
      memset(&recs[i], 0, sizeof(orc_rec));
      memcpy(recs[i].key, c->res_keys + i * nl, 8 * nl);
      recs[i].count = c->res_counts[i];
      recs[i].left = c->res_left[i];
      recs[i].right = c->res_right[i];
    }
    g_sort_nl = nl;
    qsort(recs, n, sizeof(orc_rec), rec_cmp);
    for (uint64_t i = 0; i < n; i++) {
      memcpy(c->res_keys + i * nl, recs[i].key, 8 * nl);
      c->res_counts[i] = recs[i].count;
      c->res_left[i] = recs[i].left;
      c->res_right[i] = recs[i].right;
    }
    free(recs);
  }
  c->nres = n;
  c->num_unique = total;
  c->num_purged = purged;
  c->sum_counts = sum;
  c->num_dropped = dropped;
  return 0;
}

int orc_finalize(orc_ctx *c) { return orc_finalize_ex(c, 1); }

uint64_t orc_num_results(const orc_ctx *c) { return c->nres; }

void orc_get_results(const orc_ctx *c, uint64_t *keys, uint16_t *counts, char *left, char *right) {
  memcpy(keys, c->res_keys, c->nres * 8 * c->nl);
  memcpy(counts, c->res_counts, c->nres * 2);
  memcpy(left, c->res_left, c->nres);
  memcpy(right, c->res_right, c->nres);
}

/* Every table entry before the purge (for tests of S5/S6 in isolation):
 * keys[n*nl], counts[n], exts[n*8] = left ACGT then right ACGT; sorted by key.
 * Returns the number of entries; call with NULLs to size. */
uint64_t orc_dump_table(const orc_ctx *c, uint64_t *keys, uint16_t *counts, uint16_t *exts) {
  int nl = c->nl;
  uint64_t total = 0;
  for (int r = 0; r < c->nranks; r++) total += c->tables[r].num_elems;
  if (!keys) return total;
  typedef struct { uint64_t key[ORC_MAX_LONGS]; uint16_t count; uint16_t e[8]; } full;
  full *f = (full *)malloc((total ? total : 1) * sizeof(full));
  uint64_t n = 0;
  for (int r = 0; r < c->nranks; r++) {
    const orc_table *t = &c->tables[r];
    for (uint64_t s = 0; s < t->capacity; s++) {
      const uint64_t *ks = t->keys + s * nl;
      if (ks[nl - 1] == ~0ULL) continue;
      memset(&f[n], 0, sizeof(full));
      memcpy(f[n].key, ks, 8 * nl);
      f[n].count = t->vals[s].count;
      memcpy(f[n].e, t->vals[s].left, 8);
      memcpy(f[n].e + 4, t->vals[s].right, 8);
      n++;
    }
  }
  g_sort_nl = nl;
  qsort(f, n, sizeof(full), rec_cmp); /* key is the leading member in both structs */
  for (uint64_t i = 0; i < n; i++) {
    memcpy(keys + i * nl, f[i].key, 8 * nl);
    counts[i] = f[i].count;
    memcpy(exts + i * 8, f[i].e, 16);
  }
  free(f);
  return n;
}

/* stats[]: 0 reads, 1 raw k-mers (kcount.cpp:86), 2 supermers sent, 3 k-mer
 * inserts attempted and not dropped, 4 unique before purge, 5 purged,
 * 6 results ("Total kmers", kcount.cpp:160), 7 sum of counts ("Total kmer
 * count sum", kcount_cpu.cpp:598), 8 dropped, 9 nranks, 10 nthreads, 11 table rehashes */
void orc_get_stats(const orc_ctx *c, uint64_t *stats) {
  stats[0] = c->num_reads;
  stats[1] = c->raw_kmers;
  stats[2] = c->num_supermers;
  stats[3] = c->kmers_inserted;
  stats[4] = c->num_unique;
  stats[5] = c->num_purged;
  stats[6] = c->nres;
  stats[7] = c->sum_counts;
  stats[8] = c->num_dropped;
  stats[9] = (uint64_t)c->nranks;
  stats[10] = (uint64_t)c->nthreads;
  uint64_t grows = 0;
  for (int r = 0; r < c->nranks; r++) grows += c->tables[r].num_grows;
  stats[11] = grows;
}

/* k-mer words -> ACGT string (kmer.cpp to_string), for the dump format
 * "<KMER> <count> <L> <R>" of kmer_dht.cpp:284 */
void orc_kmer_to_string(const uint64_t *kmer, int k, char *out) {
  static const char L[4] = {'A', 'C', 'G', 'T'};
  for (int i = 0; i < k; i++) out[i] = L[(kmer[i / 32] >> (2 * (31 - (i % 32)))) & 3];
  out[k] = 0;
}
