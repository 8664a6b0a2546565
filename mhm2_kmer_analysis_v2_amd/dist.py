"""Shard exchange for the multi-GPU flow: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Replaces the reference's only bulk exchange, the aggregated UPC++ RPC of
ThreeTierAggrStore::update (src/kcount/kmer_dht.cpp:143-151,247-258), inside the
node.  Two flows, one protocol (sizes first, then one group of point-to-point transfers):

  single pass (ShardedKmerAnalysis.single_pass; what bench.py runs when a shard's share of the regions holds its k-mers):
      a shard owns level-1 BUCKETS (csrc/kc_shard.hpp).
      Per block of reads every rank runs its ordinary level-1 pass and packs what other shards own into one wire
      segment per destination (kc_shard_extract); the ranks swap the segment sizes (an N x N all-to-all of int64), then
      the segments, each landing in memory of the receiving context (kc_shard_reserve) where level 2 later reads it in
      place (kc_shard_commit).  A rank's own share never moves.
  records (what bench.py runs for full-size shards): records binned by owner (kc_extract_partition), inserted by the
      receiver's level 1 (kc_insert_records): two more passes over every record, but every shard uses the whole geometry.
      With wire units (KmerCounter(wire_units=True), csrc/kc_wire6.hpp) the records are the library's own -- at k=21 units
      of four six-byte records of the mixed k-mer -- and a destination's records come in `pieces` pieces by the top bits
      of their level-1 bucket: a transfer per piece, what arrives laid piece 0 of every sender first, then piece 1, ...

Nothing else is communicated: ownership is a pure function of the k-mer, finalize is per shard.
"""
import torch
import torch.distributed as dist


def exchange_counts(send_counts, group=None):
    """send_counts: int64 tensor [world * pieces] (units of every piece, destination after destination).  Returns what
    every sender has for this rank, sender after sender."""
    recv = torch.empty_like(send_counts)
    dist.all_to_all_single(recv, send_counts, group=group)
    return recv


def start_exchange(send, send_counts, recv_counts, seg_capacity, num_longs, recv=None, group=None, reserve=None, pieces=1):
    """Begin the all-to-all-v of one block.  send: int64 tensor laid out as world * pieces pieces of seg_capacity units
    (num_longs words each), the first send_counts[d * pieces + q] units of piece q of destination d being valid
    (pieces > 1: the wire units of kc_wire_unit, whose sender keeps a destination's records apart by the top bits of their
    level-1 bucket).  Returns (works, recv tensor, n_received, parts): what the OTHER ranks sent, once every work has been
    waited for; parts = [(first word, words)] per received piece.  Without `reserve` the parts are packed back to back in
    `recv` (grown as needed), piece 0 of every sender first, then piece 1, ...: what lies side by side then holds the
    same buckets, which is what the receiver's level 1 wants; with `reserve` (the single-pass flow: reserve(nwords) ->
    tensor inside the receiving context) every part starts on a 16-byte boundary of a fresh reservation.  The rank's own
    share stays where it is: local_share(...) is its view."""
    world = dist.get_world_size(group)
    me = dist.get_rank(group)
    sc = [int(x) for x in send_counts.tolist()]
    rc = [int(x) for x in recv_counts.tolist()]
    align = 2 if reserve is not None else 1
    starts, pos = {}, 0
    for q in range(pieces):
        for d in range(world):
            if d != me:
                starts[(d, q)] = pos
                pos += (rc[d * pieces + q] * num_longs + align - 1) // align * align
    total = sum(rc) - sum(rc[me * pieces:(me + 1) * pieces])  # what arrives from the other ranks
    if reserve is not None:
        recv = reserve(pos)
    elif recv is None or recv.numel() < max(pos, 1):
        recv = torch.empty(max(pos, 1), dtype=send.dtype, device=send.device)
    # all-to-all-v as one group of point-to-point transfers (RCCL: a single ncclGroup of
    # ncclSend/ncclRecv over xGMI; gloo: the same ops over TCP).  The segments of `send` are not
    # contiguous, so a single-buffer all_to_all_single would need an extra packing pass over HBM.
    ops, parts = [], []
    for d in range(world):
        if d == me:
            continue  # this rank's own share never travels
        peer = dist.get_global_rank(group, d) if group is not None else d
        for q in range(pieces):  # (both sides post a peer's pieces in the same order)
            j = d * pieces + q
            base = j * seg_capacity * num_longs
            src = send[base:base + sc[j] * num_longs]
            dst = recv[starts[(d, q)]:starts[(d, q)] + rc[j] * num_longs]
            if rc[j]:
                ops.append(dist.P2POp(dist.irecv, dst, peer, group))
                parts.append((starts[(d, q)], rc[j] * num_longs))
            if sc[j]:
                ops.append(dist.P2POp(dist.isend, src, peer, group))
    works = dist.batch_isend_irecv(ops) if ops else []
    return works, recv, total, parts


def local_share(send, send_counts, seg_capacity, num_longs, group=None, pieces=1):
    """The records this rank keeps for itself: views of its own pieces of `send` (each from its first word to the end of the
    buffer) with their units, piece after piece, and the units in all."""
    me = dist.get_rank(group)
    own = []
    for q in range(pieces):
        j = me * pieces + q
        own.append((send[j * seg_capacity * num_longs:], int(send_counts[j])))
    return own, sum(n for _, n in own)


def exchange_records(send, send_counts, recv_counts, seg_capacity, num_longs, recv=None, group=None, pieces=1):
    """Blocking form of start_exchange: returns (received from the others, their number, own pieces [(view, units)], their units)."""
    works, recv, total, _ = start_exchange(send, send_counts, recv_counts, seg_capacity, num_longs, recv, group, pieces=pieces)
    for w in works:
        w.wait()
    own, n_own = local_share(send, send_counts, seg_capacity, num_longs, group, pieces)
    return recv, total, own, n_own


class ShardedKmerAnalysis:
    """count_kmers + flush_updates + finish_updates (src/kcount/kcount.cpp:54-104,142-161) across the ranks
    of one node.

    Records flow: `extract(block, send, seg_capacity) -> counts` and `insert(recv, n)` are the two device entry points
    (KmerCounter.extract_partition / insert_records).  Single-pass flow (`reserve` given, see single_pass()):
    `extract(block, send, seg_words) -> words per destination` (KmerCounter.shard_extract), `reserve(nwords) -> tensor`
    (KmerCounter.shard_reserve) and `insert(segment, nwords)` (KmerCounter.shard_commit), num_longs = 1.  The CPU tests
    plug in stand-ins, so that the exchange logic itself is what they exercise.

    Two send buffers (and, records flow, two receive buffers): while block i travels (RCCL's own stream), block i-1 joins
    the table and block i+1 is extracted on the compute stream.

    Stream rule (GPU): a finished RCCL work orders only torch's CURRENT stream behind the transfer, so the library's
    kernels must run on that stream or they could read `recv` before it has landed.  Pass the KmerCounter as
    `counter` and this class puts it on torch's current stream of `device` (kc_set_stream); without one the caller
    must have done so itself.  The C++ twin (csrc/kc_exchange.hpp) orders its two streams with events instead."""

    def __init__(self, extract, insert, num_longs, seg_capacity, device, group=None, counter=None, reserve=None, pieces=1, insert_pieces=None):
        self.extract, self.insert, self.reserve = extract, insert, reserve
        # insert_pieces(first piece, stride in units, [units]): the rank's own pieces of the send buffer in one call
        # (KmerCounter.insert_record_pieces); without it they are inserted one by one
        self.insert_pieces = insert_pieces
        self.nl, self.seg, self.pieces = num_longs, seg_capacity, pieces
        self.group = group
        self.world = dist.get_world_size(group)
        self.device = device
        if counter is not None and torch.device(device).type == "cuda":
            stream = torch.cuda.current_stream(device)
            if stream.cuda_stream == 0:
                # the null stream reads as "use your own" to the library: give both a real one
                stream = torch.cuda.Stream(device=device)
                torch.cuda.set_stream(stream)
            counter.set_stream(stream.cuda_stream)
            self._stream = stream
        self.send = [torch.zeros(self.world * pieces * seg_capacity * num_longs, dtype=torch.int64, device=device) for _ in range(2)]
        self.recv = [None, None]
        self.pending = None
        self.i = 0
        self.sent = 0
        self.received = 0

    @classmethod
    def single_pass(cls, counter, extract, seg_words, device, group=None):
        """The single-pass flow over a KmerCounter: extract(block, send, seg_words) must call counter.shard_extract."""
        return cls(extract, counter.shard_commit, 1, seg_words, device, group, counter, reserve=counter.shard_reserve)

    def _complete(self):
        if self.pending is None:
            return 0
        works, recv, n, own, n_own, pieces = self.pending
        self.pending = None
        if own and self.insert_pieces is not None:  # straight from the send buffer (reused two blocks later at the earliest)
            self.insert_pieces(own[0][0], self.seg, [u for _, u in own])
        else:
            for piece, units in own or []:
                if units:
                    self.insert(piece[:units * self.nl], units)
        for w in works:
            w.wait()
        if self.reserve is not None:
            for first, words in pieces:  # every received segment joins its buckets where it lies
                self.insert(recv[first:first + words], words)
        elif n:
            self.insert(recv, n)
        self.received += n + n_own
        return n + n_own

    def add_block(self, block):
        b = self.i % 2
        self.i += 1
        # send[b] is free: its last transfer (two blocks ago) was waited for when the block after it was added
        # A rank whose extraction fails (a segment too small, a bad character) must not leave the others waiting in the
        # exchange of the sizes: it takes part with sizes no block can have, and every rank raises together.
        failure = None
        try:
            counts = self.extract(block, self.send[b], self.seg)
        except Exception as e:  # noqa: BLE001 -- whatever it is, it is re-raised below, after the collective
            failure, counts = e, [-1] * (self.world * self.pieces)
        sc_host = torch.as_tensor([int(c) for c in counts], dtype=torch.int64)
        rc = exchange_counts(sc_host.to(self.device), self.group).cpu()  # the one host round trip of a block
        if failure is not None:
            raise failure
        if bool((rc < 0).any()):
            raise RuntimeError("rank(s) %s failed to extract their block" % [int(i) for i in torch.nonzero(rc < 0).flatten()])
        works, recv, n, pieces = start_exchange(self.send[b], sc_host, rc, self.seg, self.nl, self.recv[b], self.group, self.reserve, self.pieces)
        if self.reserve is None:
            self.recv[b] = recv
        own, n_own = (None, 0) if self.reserve is not None else local_share(self.send[b], sc_host, self.seg, self.nl, self.group, self.pieces)
        self._complete()  # the previous block: its transfer has had this block's extraction to finish
        self.pending = (works, recv, n, own, n_own, pieces)
        self.sent += int(sc_host.sum())
        return n + n_own

    def finish(self):
        """Wait for and insert the last block in flight (call before finalize)."""
        return self._complete()
