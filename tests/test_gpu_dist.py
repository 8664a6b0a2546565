"""dist.py on the GPU: ShardedKmerAnalysis over the nccl backend (= RCCL) with the real device entry points
(kc_extract_partition / kc_insert_records, and kc_shard_* for the single-pass flow).  The box has one GPU, so the communicator has one member -- the N > 1
code path end to end (counts all-to-all, grouped send/recv, double buffering, the stream rule) minus the wire; two
ranks over gloo are covered on the CPU by test_dist_gloo.py."""
import os

import numpy as np
import pytest

import mhm2_kmer_analysis_v2_amd as pkg
from helpers import random_reads
from oracle import cpu_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture
def nccl_world1():
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    import torch
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("flow", ["records", "single-pass"])
@pytest.mark.parametrize("k,tuning", [(21, dict(p1=1024, p2=1024)), (51, None)], ids=["k21-compact", "k51"])
def test_sharded_analysis_over_nccl(nccl_world1, k, tuning, flow):
    import torch
    from mhm2_kmer_analysis_v2_amd.dist import ShardedKmerAnalysis
    rng = np.random.default_rng(31 + k)
    reads, quals = random_reads(rng, 4000, min_len=k - 2, max_len=k + 120, genome_len=6000)
    b, q, offs = O.reads_to_arrays(reads, quals)
    o = O.Oracle(k, nranks=4, nthreads=4)
    o.add_reads(b, q, offs)
    want = o.finalize()
    wst = o.stats()
    o.close()
    dev = torch.device("cuda", 0)
    nl = pkg.lib().kc_record_longs(k)
    db = torch.from_numpy(b).to(dev)
    dq = torch.from_numpy(q).to(dev)
    do = torch.from_numpy(offs.astype(np.int64)).to(dev)
    with pkg.KmerCounter(k, rank_me=0, rank_n=1, tuning=tuning) as kc:  # created on its own stream: the class must move it
        def extract(block, send, seg_cap):
            r0, r1 = block
            o0 = int(offs[r0])
            return kc.extract_partition(db[o0:], dq[o0:], do[r0:r1 + 1] - do[r0], send, seg_cap, nreads=r1 - r0)

        def shard_extract(block, send, seg_words):
            r0, r1 = block
            o0 = int(offs[r0])
            return kc.shard_extract(db[o0:], dq[o0:], do[r0:r1 + 1] - do[r0], send, seg_words, nreads=r1 - r0)

        if flow == "records":
            sh = ShardedKmerAnalysis(extract, lambda recv, n: kc.insert_records(recv, n), nl, 600 * 150, dev, counter=kc)
        else:  # kc_shard_extract / kc_shard_reserve / kc_shard_commit: with one member nothing leaves the shard
            sh = ShardedKmerAnalysis.single_pass(kc, shard_extract, 600 * 150 * nl + 2048, dev)
        for r0 in range(0, 4000, 600):  # seven blocks: both buffers are reused several times
            sh.add_block((r0, min(4000, r0 + 600)))
        sh.finish()
        got = kc.sorted_results()
        st = kc.stats()
    assert sh.sent == sh.received == (wst["kmers_inserted"] if flow == "records" else 0)
    assert st["kmers_inserted"] == wst["kmers_inserted"]
    for g, w in zip(got, want):
        assert g.shape == w.shape and (g == w).all()
    assert st["num_unique"] == wst["unique"]
