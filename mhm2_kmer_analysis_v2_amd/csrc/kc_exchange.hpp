// kc_exchange.hpp -- the intra-node shard exchange in C++ over RCCL (xGMI), for a C++/UPC++ host such as MHM2.
//
// Replaces the reference's only bulk exchange inside the node: the aggregated RPC of
// ThreeTierAggrStore<Supermer>::update / flush_updates (src/kcount/kmer_dht.cpp:143-151,247-258) that carries every
// supermer to the rank owning its k-mers.  One process per GPU, one ShardExchange per process.  Two flows:
//
//   BUCKETS (default): the single-pass shard flow of csrc/kc_shard.hpp -- a shard owns level-1 buckets
//     per block of reads   kc_shard_extract              the block through level 1; what other shards own packed into
//                                                        one wire segment per destination (compute stream)
//                          ncclAllGather of the sizes    N x N u64: everybody learns what it will receive
//                          kc_shard_reserve              room for it inside the receiving context (it stays there)
//                          grouped ncclSend / ncclRecv   all-to-all-v of the segments on a SIDE stream, so that it
//                                                        overlaps the extraction of the next block
//                          kc_shard_commit               per received segment, once the side stream says it has landed
//                                                        (an event, no host wait on the transfer): read in place by level 2
//     the rank's own share never moves at all.
//   RECORDS: records binned by owner (kc_extract_partition), inserted by the receiver's level 1 (kc_insert_records) -- two
//     more passes over every record, but every shard uses the whole geometry (a full-size shard at any N), it also works
//     for a context on the global-table path, and its records are what a upcxx::rpc would carry across nodes.  With
//     contexts created with KC_FLAG_WIRE_UNITS the records are the library's own (csrc/kc_wire6.hpp: at k=21 units of four
//     six-byte records of the mixed k-mer, a destination's records in `pieces` pieces by the top bits of their level-1
//     bucket): a send / receive per piece, what arrives laid piece 0 of every sender first, then piece 1, ..., the rank's
//     own pieces handed over where they lie (kc_insert_record_pieces).
//
// Two send buffers alternate, so block i travels while block i+1 is extracted.  Nothing else is communicated: ownership
// is a pure function of the k-mer (kc_shard_owner / kc_owner), finalize is per shard.
// xGMI is point to point: every rank ships 1/N of its records to each peer over that peer's own link (76.8 GB/s a
// direction at its peak).  At the rate one GPU's kernels run (50 M reads = 6.4 G records in ~72 ms) the links are what
// bounds an 8-shard stage if a record travels as its 8-byte word: 7/8 x 6.4 G x 8 B = 44.8 GB per step and shard = 6.4 GB
// per link = 83 ms.  So the BUCKETS flow ships the compact records of k=21 as FIVE bytes (kc_shard.hpp,
// SHARD_WIRE_COMPACT: 4.0 GB per link = 52 ms, under the kernels' time and overlapped with it on the side stream), the
// RECORDS flow in wire units SIX (4.8 GB per link = 62 ms under its 86 ms of kernels); longer k-mers ship whole words
// (DESIGN.md section 6 has the link budget for N = 2, 4, 8).
//
// Header-only over the C ABI (include/kcount_mi355.h) and <rccl/rccl.h>; no exceptions (MHM2 calls this inside UPC++
// progress): every method returns a KC_* status, last_error() has the text.  The Python twin used by bench.py and the
// tests is mhm2_kmer_analysis_v2_amd/dist.py (torch.distributed, same protocol).
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/kcount_mi355.h"

namespace kcount_mi355 {

class ShardExchange {
 public:
  enum Flow { BUCKETS = 0, RECORDS = 1 };

 private:
  kc_ctx *ctx;
  ncclComm_t comm;
  int me, n, nl;
  int Q = 1;                       // RECORDS: pieces per destination (kc_wire_unit), 1 for k-mer records
  uint64_t seg;                    // BUCKETS: words, RECORDS: records per destination segment of a send buffer
  Flow flow;
  hipStream_t compute = nullptr;   // the context's stream (ours unless the caller gave one)
  hipStream_t side = nullptr;      // RCCL's stream
  bool own_compute = false;
  uint64_t *send[2] = {nullptr, nullptr}, *recv[2] = {nullptr, nullptr};  // recv[]: RECORDS only
  uint64_t recv_cap[2] = {0, 0};
  uint64_t *d_counts = nullptr, *d_all = nullptr;  // this rank's N sizes; everybody's N x N
  uint64_t *h_all = nullptr;                       // pinned
  hipEvent_t arrived[2] = {nullptr, nullptr};      // side stream: the block has landed
  hipEvent_t consumed[2] = {nullptr, nullptr};     // compute stream: whatever read send[i] / recv[i] is done
  hipEvent_t extracted = nullptr;                  // compute stream: the block's segments are complete
  bool used[2] = {false, false};
  struct Piece { const uint64_t *p; uint64_t words; };
  struct Pending {
    bool any = false;
    int buf = 0;
    uint64_t n_own = 0, n_recv = 0;  // RECORDS
    std::vector<uint64_t> own;       // RECORDS: units of this rank's own pieces
    std::vector<Piece> pieces;       // BUCKETS: the received segments, where they landed
  } pending;
  uint64_t blocks = 0, sent = 0, received = 0;
  std::string err;

  int fail(int status, const char *what, const char *detail) {
    err = std::string(what) + ": " + detail;
    return status;
  }
#define KCX_HIP(call)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) return fail(KC_ERR_HIP, #call, hipGetErrorString(e_));       \
  } while (0)
#define KCX_NCCL(call)                                                                 \
  do {                                                                                 \
    ncclResult_t r_ = (call);                                                          \
    if (r_ != ncclSuccess) return fail(KC_ERR_HIP, #call, ncclGetErrorString(r_));     \
  } while (0)
#define KCX_KC(call)                                                                   \
  do {                                                                                 \
    int s_ = (call);                                                                   \
    if (s_ != KC_OK) return fail(s_, #call, kc_last_error());                          \
  } while (0)

  static constexpr uint64_t POISON = ~0ULL;  // a size that says "this rank failed" in the exchange of the sizes

  // the previous block: what it received joins the table (compute stream, ordered behind the transfer by an event)
  int complete() {
    if (!pending.any) return KC_OK;
    const int b = pending.buf;
    if (flow == BUCKETS) {
      if (!pending.pieces.empty()) KCX_HIP(hipStreamWaitEvent(compute, arrived[b], 0));
      for (const Piece &pc : pending.pieces) {
        KCX_KC(kc_shard_commit(ctx, pc.p, pc.words));
        received += pc.words;
      }
      pending.pieces.clear();
    } else {
      // this rank's own pieces, straight from the send buffer, in one call
      if (pending.n_own) KCX_KC(kc_insert_record_pieces(ctx, send[b] + (uint64_t)me * Q * seg * nl, seg, Q, pending.own.data()));
      // always: the extraction of two blocks on overwrites send[b] on this stream, and the side stream's ncclSend of
      // this block may still be reading it -- also when this rank received nothing
      KCX_HIP(hipStreamWaitEvent(compute, arrived[b], 0));
      if (pending.n_recv) KCX_KC(kc_insert_records(ctx, recv[b], pending.n_recv));
      KCX_HIP(hipEventRecord(consumed[b], compute));
      received += pending.n_own + pending.n_recv;
    }
    pending.any = false;
    return KC_OK;
  }

 public:
  // ctx: this rank's context (created with the same rank_me / rank_n as `comm`, and -- BUCKETS -- the same sizes and
  // tuning on every rank); num_longs: kc_record_longs(k) -- for RECORDS with contexts created with KC_FLAG_WIRE_UNITS the
  // words of a unit (kc_wire_unit), and seg_capacity in units --; seg_capacity: what one block may send to one shard, in WORDS
  // for BUCKETS (a block of R reads of length L needs about R * (L - k - 1) / rank_n * num_longs * 1.1 + 1024), in
  // RECORDS for RECORDS (R * (L - k - 1) / rank_n * 1.25; with wire units: units per PIECE, and `pieces` the pieces per
  // destination, both from kc_wire_unit); compute_stream: the stream the context's kernels should run
  // on, NULL = a stream of this object's own.
  ShardExchange(kc_ctx *ctx_, ncclComm_t comm_, int rank_me, int rank_n, int num_longs, uint64_t seg_capacity,
                hipStream_t compute_stream = nullptr, Flow flow_ = BUCKETS, int pieces = 1)
      : ctx(ctx_), comm(comm_), me(rank_me), n(rank_n), nl(num_longs), Q(flow_ == RECORDS ? pieces : 1), seg(seg_capacity), flow(flow_),
        compute(compute_stream) {}
  ShardExchange(const ShardExchange &) = delete;
  ShardExchange &operator=(const ShardExchange &) = delete;

  int init() {
    if (!ctx || n < 1 || me < 0 || me >= n || nl < 1 || !seg || Q < 1) return fail(KC_ERR_INVALID_ARG, "ShardExchange", "bad arguments");
    if (!compute) {
      KCX_HIP(hipStreamCreateWithFlags(&compute, hipStreamNonBlocking));
      own_compute = true;
    }
    KCX_KC(kc_set_stream(ctx, (void *)compute));
    KCX_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    for (int b = 0; b < 2; b++) {
      KCX_HIP(hipMalloc((void **)&send[b], (size_t)n * Q * seg * (flow == BUCKETS ? 1 : nl) * 8));
      KCX_HIP(hipEventCreateWithFlags(&arrived[b], hipEventDisableTiming));
      KCX_HIP(hipEventCreateWithFlags(&consumed[b], hipEventDisableTiming));
    }
    KCX_HIP(hipEventCreateWithFlags(&extracted, hipEventDisableTiming));
    KCX_HIP(hipMalloc((void **)&d_counts, (size_t)n * Q * 8));
    KCX_HIP(hipMalloc((void **)&d_all, (size_t)n * n * Q * 8));
    KCX_HIP(hipHostMalloc((void **)&h_all, (size_t)n * n * Q * 8, hipHostMallocDefault));
    return KC_OK;
  }

  ~ShardExchange() {
    if (side) (void)hipStreamSynchronize(side);
    if (compute) (void)hipStreamSynchronize(compute);
    for (int b = 0; b < 2; b++) {
      if (send[b]) (void)hipFree(send[b]);
      if (recv[b]) (void)hipFree(recv[b]);
      if (arrived[b]) (void)hipEventDestroy(arrived[b]);
      if (consumed[b]) (void)hipEventDestroy(consumed[b]);
    }
    if (extracted) (void)hipEventDestroy(extracted);
    if (d_counts) (void)hipFree(d_counts);
    if (d_all) (void)hipFree(d_all);
    if (h_all) (void)hipHostFree(h_all);
    if (side) (void)hipStreamDestroy(side);
    if (own_compute && compute) {
      (void)kc_set_stream(ctx, nullptr);
      (void)hipStreamDestroy(compute);
    }
  }

  // count_kmers' loop body for one block of reads (src/kcount/kcount.cpp:71-90 + kmer_dht.cpp:247-250): extract, ship.
  // bases / quals / offsets as for kc_submit_reads.  Collective: every rank calls it the same number of times (a rank
  // that has run out of reads calls it with nreads = 0).
  int add_block(const uint8_t *bases, const uint8_t *quals, const uint64_t *offsets, uint64_t nreads, int on_device) {
    const int b = (int)(blocks & 1);
    blocks++;
    // send[b] was last read by the transfers of two blocks ago (BUCKETS) or by the inserts of its own share (RECORDS)
    if (used[b]) {
      if (flow == BUCKETS) KCX_HIP(hipEventSynchronize(arrived[b]));  // the kernels below overwrite it at once
      // RECORDS: the extraction is ordered behind those inserts by the compute stream itself
    }
    const size_t NQ = (size_t)n * Q;  // sizes per rank: one per destination (BUCKETS), one per piece of every destination (RECORDS)
    std::vector<uint64_t> counts(NQ, 0);
    // A rank whose extraction fails (a segment too small, a bad character) must not leave the others waiting in the
    // exchange of the sizes: it takes part with a size no block can have, and every rank returns an error together.
    int local_rc = KC_OK;
    if (flow == BUCKETS) {
      local_rc = kc_shard_extract(ctx, bases, quals, offsets, nreads, on_device, send[b], seg, counts.data());
    } else if (nreads) {
      local_rc = kc_extract_partition(ctx, bases, quals, offsets, nreads, on_device, send[b], seg, counts.data());
    }
    if (local_rc != KC_OK) {
      fail(local_rc, flow == BUCKETS ? "kc_shard_extract" : "kc_extract_partition", kc_last_error());
      std::fill(counts.begin(), counts.end(), POISON);
    }
    // kc_shard_extract / kc_extract_partition return with the segments complete on the compute stream (they synchronize
    // it); the side stream is ordered behind the compute stream all the same, so that the transfers below never depend
    // on that detail
    KCX_HIP(hipEventRecord(extracted, compute));
    KCX_HIP(hipStreamWaitEvent(side, extracted, 0));
    // everybody's sizes: N x N, row s = what rank s sends to each shard
    KCX_HIP(hipMemcpyAsync(d_counts, counts.data(), NQ * 8, hipMemcpyHostToDevice, side));
    KCX_NCCL(ncclAllGather(d_counts, d_all, NQ, ncclUint64, comm, side));
    KCX_HIP(hipMemcpyAsync(h_all, d_all, (size_t)n * NQ * 8, hipMemcpyDeviceToHost, side));
    KCX_HIP(hipStreamSynchronize(side));
    if (local_rc != KC_OK) return local_rc;
    for (size_t i = 0; i < (size_t)n * NQ; i++)
      if (h_all[i] == POISON) return fail(KC_ERR_STATE, "ShardExchange::add_block", "another rank failed to extract its block");
    const uint64_t unit = flow == BUCKETS ? 1 : (uint64_t)nl;  // words per counted thing
    uint64_t total = 0;  // BUCKETS: every sender's part starts on a 16-byte boundary
    for (int s = 0; s < n; s++) {
      if (s == me) continue;
      for (int q = 0; q < Q; q++) {
        const uint64_t w = h_all[(size_t)s * NQ + (size_t)me * Q + q];
        total += flow == BUCKETS ? ((w + 1) & ~1ULL) : w;
      }
    }
    uint64_t *dst = nullptr;
    if (flow == BUCKETS) {
      KCX_KC(kc_shard_reserve(ctx, total, &dst));  // inside the context, for good: level 2 reads it in place
    } else {
      if (total > recv_cap[b]) {
        if (used[b]) KCX_HIP(hipEventSynchronize(consumed[b]));
        if (recv[b]) KCX_HIP(hipFree(recv[b]));
        recv[b] = nullptr;
        recv_cap[b] = total + total / 8 + 1024;
        KCX_HIP(hipMalloc((void **)&recv[b], (size_t)recv_cap[b] * nl * 8));
      }
      dst = recv[b];
    }
    // the previous block: its transfer has had this block's extraction to finish
    int rc = complete();
    if (rc) return rc;
    // all-to-all-v: one group of point-to-point transfers (each pair has its own xGMI link)
    if (flow == RECORDS && used[b]) KCX_HIP(hipStreamWaitEvent(side, consumed[b], 0));
    KCX_NCCL(ncclGroupStart());
    // where every piece lands: piece 0 of every sender first, then piece 1, ... -- what lies side by side then holds the
    // same level-1 buckets, which is what the receiver's level 1 wants (kc_wire6.hpp)
    std::vector<uint64_t> at(NQ, 0);
    uint64_t pos = 0;
    for (int q = 0; q < Q; q++)
      for (int d = 0; d < n; d++) {
        if (d == me) continue;
        const uint64_t rcv = h_all[(size_t)d * NQ + (size_t)me * Q + q];
        at[(size_t)d * Q + q] = pos;
        pos += flow == BUCKETS ? ((rcv + 1) & ~1ULL) : rcv;
      }
    for (int d = 0; d < n; d++) {
      if (d == me) continue;  // this rank's own share never travels
      for (int q = 0; q < Q; q++) {  // (both sides post a peer's pieces in the same order)
        const size_t j = (size_t)d * Q + q;
        const uint64_t sc = counts[j], rcv = h_all[(size_t)d * NQ + (size_t)me * Q + q];
        if (rcv) {
          KCX_NCCL(ncclRecv(dst + at[j] * unit, (size_t)rcv * unit, ncclUint64, d, comm, side));
          if (flow == BUCKETS) pending.pieces.push_back(Piece{dst + at[j], rcv});
        }
        if (sc) KCX_NCCL(ncclSend(send[b] + (uint64_t)j * seg * unit, (size_t)sc * unit, ncclUint64, d, comm, side));
        sent += sc;
      }
    }
    KCX_NCCL(ncclGroupEnd());
    KCX_HIP(hipEventRecord(arrived[b], side));
    used[b] = true;
    pending.any = true;
    pending.buf = b;
    if (flow == RECORDS) {
      pending.own.assign(counts.begin() + (size_t)me * Q, counts.begin() + (size_t)(me + 1) * Q);
      pending.n_own = 0;
      for (uint64_t v : pending.own) pending.n_own += v;
      sent += pending.n_own;
      pending.n_recv = total;
    }
    return KC_OK;
  }

  // KmerDHT::flush_updates (kmer_dht.cpp:252-258): the last block in flight joins the table; call before kc_finalize
  int finish() {
    int rc = complete();
    if (rc) return rc;
    KCX_KC(kc_flush(ctx));
    return KC_OK;
  }

  // BUCKETS: words shipped to / received from other ranks; RECORDS: records, the rank's own share included
  uint64_t sent_units() const { return sent; }
  uint64_t received_units() const { return received; }
  uint64_t records_sent() const { return sent; }
  uint64_t records_received() const { return received; }
  const char *last_error() const { return err.c_str(); }
  hipStream_t compute_stream() const { return compute; }
#undef KCX_HIP
#undef KCX_NCCL
#undef KCX_KC
};

}  // namespace kcount_mi355
