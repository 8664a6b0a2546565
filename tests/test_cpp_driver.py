"""The C++ adapters of kcount_driver.hpp (the reference's ParseAndPackGPUDriver / HashTableGPUDriver
surface re-created over the C ABI): they compile everywhere, and on a GPU a small C++ program driven
like src/kcount/kcount_gpu.cpp reproduces the oracle's dump."""
import os
import subprocess

import numpy as np
import pytest

from helpers import random_reads
from oracle import cpu_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mhm2_kmer_analysis_v2_amd", "csrc")
SRC = os.path.join(ROOT, "tests", "cpp", "test_driver.cpp")
HIPCC = "/opt/rocm/bin/hipcc"


def build(tmp):
    exe = os.path.join(str(tmp), "test_driver")
    subprocess.check_call([HIPCC, "-std=c++17", "-O1", "-o", exe, SRC, "-L" + CSRC, "-lkcount_mi355", "-Wl,-rpath," + CSRC])
    return exe


def test_adapters_compile_and_link(tmp_path):
    assert os.path.exists(build(tmp_path))


def build_exchange(tmp):
    exe = os.path.join(str(tmp), "test_exchange")
    subprocess.check_call([HIPCC, "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_exchange.cpp"), "-L" + CSRC,
                           "-lkcount_mi355", "-lrccl", "-Wl,-rpath," + CSRC])
    return exe


def test_rccl_exchange_compiles_and_links(tmp_path):
    """kc_exchange.hpp (ShardExchange over rccl.h: all-gathered counts, grouped ncclSend/ncclRecv on a side stream)"""
    assert os.path.exists(build_exchange(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("flow", ["buckets", "records", "wire-units"])
@pytest.mark.parametrize("k", [21, 51])
def test_rccl_exchange_one_rank_matches_oracle(tmp_path, k, flow):
    """the C++ exchange driven like count_kmers + KmerDHT, one-member communicator (the box has one GPU): blocks,
    sizes through RCCL, the receiving side ordered by events; result = the oracle's dump.  Both flows of the class: the
    single-pass one (a shard owns level-1 buckets), the records one (hash ownership) and the records one in wire units
    (csrc/kc_wire6.hpp: six-byte records at k = 21, k-mer records at k = 51)."""
    exe = build_exchange(tmp_path)
    env = dict(os.environ, KC_EXCHANGE_FLOW=flow)
    rng = np.random.default_rng(78)
    reads, quals = random_reads(rng, 700, min_len=25, max_len=160, genome_len=2500, n_rate=0.0)
    masked = ["".join(c.lower() if ord(x) < 33 + 20 else c for c, x in zip(r, q)) for r, q in zip(reads, quals)]
    out = subprocess.run([exe, str(k)], input="\n".join(masked) + "\n", capture_output=True, text=True, check=True, env=env).stdout
    got = [l[5:] for l in out.splitlines() if l.startswith("KMER ")]
    (keys, counts, left, right), st = O.count_reads(reads, quals, k=k)
    want = sorted("%s %d %s %s" % (O.kmer_to_string(keys[i], k), counts[i], chr(left[i]), chr(right[i])) for i in range(len(counts)))
    assert got == want and len(want) > 50


def test_compact_record_mix_is_a_bijection(tmp_path):
    """kc_feistel_fwd / kc_feistel_inv (host build of kc_common.hpp): a permutation with the stated inverse, evenly
    spread bucket / region / slot bits (tests/cpp/test_mix.cpp)."""
    exe = os.path.join(str(tmp_path), "test_mix")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_mix.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "bad=0", out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["records", "ascii", "packed", "wire"])
def test_cpp_driver_matches_oracle(tmp_path, mode):
    k = 21
    exe = build(tmp_path)
    rng = np.random.default_rng(77)
    reads, quals = random_reads(rng, 400, min_len=25, max_len=160, genome_len=1500)
    masked = ["".join(c.lower() if ord(x) < 33 + 20 else c for c, x in zip(r, q)) for r, q in zip(reads, quals)]
    out = subprocess.run([exe, str(k), mode], input="\n".join(masked) + "\n", capture_output=True, text=True, check=True).stdout
    (keys, counts, left, right), st = O.count_reads(reads, quals, k=k)
    want = sorted("%s %d %s %s" % (O.kmer_to_string(keys[i], k), counts[i], chr(left[i]), chr(right[i])) for i in range(len(counts)))
    assert out.splitlines() == want and len(want) > 50


@pytest.mark.gpu
@pytest.mark.parametrize("k", [21, 31])
def test_cpp_driver_contig_pass_matches_oracle(tmp_path, k):
    """HashTableDriver in the contig pass, driven as kmer_dht.cpp:158-171 drives the reference's: reads, flush,
    init_ctg_kmers, every contig as one packed supermer with its depth as the count, done_ctg_kmer_inserts,
    done_all_inserts -- against the oracle's insert_supermer_from_ctg (kcount_cpu.cpp:357-407)."""
    exe = build(tmp_path)
    rng = np.random.default_rng(79 + k)
    genome = "".join(rng.choice(list("ACGT"), size=2500))
    reads, quals = [], []
    for _ in range(500):
        a = int(rng.integers(0, len(genome) - 160))
        ln = int(rng.integers(k + 2, 150))
        s = list(genome[a:a + ln])
        for j in range(ln):
            if rng.random() < 0.01:
                s[j] = "ACGT"[int(rng.integers(0, 4))]
        reads.append("".join(s))
        quals.append("".join("I" if rng.random() > 0.03 else "#" for _ in range(ln)))
    ctgs, depths = [], []
    for i in range(30):
        a = int(rng.integers(0, len(genome) - 400))
        ctgs.append(genome[a:a + int(rng.integers(k + 2, 300))])
        depths.append(int(rng.integers(1, 50)))
    ctgs += [ctgs[0], ctgs[1][:len(ctgs[1]) // 2], "".join(rng.choice(list("ACGT"), size=300))]
    depths += [depths[0] + 3, 2, 13]
    masked = ["".join(c.lower() if ord(x) < 33 + 20 else c for c, x in zip(r, q)) for r, q in zip(reads, quals)]
    text = "\n".join(masked) + "\n" + "".join(">%d %s\n" % (d, c) for c, d in zip(ctgs, depths))
    out = subprocess.run([exe, str(k), "ctg"], input=text, capture_output=True, text=True, check=True).stdout
    o = O.Oracle(k, nranks=1, nthreads=1)
    b, q, offs = O.reads_to_arrays(reads, quals)
    o.add_reads(b, q, offs)
    for c, d in zip(ctgs, depths):
        o.add_ctg(c, d)
    keys, counts, left, right = o.finalize()
    o.close()
    want = sorted("%s %d %s %s" % (O.kmer_to_string(keys[i], k), counts[i], chr(left[i]), chr(right[i])) for i in range(len(counts)))
    plain = O.count_reads(reads, quals, k=k)[0]
    assert out.splitlines() == want and len(want) > len(plain[1])  # (the contigs added k-mers the reads alone do not keep)


SURFACE = os.path.join(ROOT, "tests", "cpp", "test_surface.cpp")
REFERENCE = "/root/reference/src"


def build_surface(tmp):
    exe = os.path.join(str(tmp), "test_surface")
    subprocess.check_call([HIPCC, "-std=c++17", "-O1", "-Wall", "-Werror", "-o", exe, SURFACE, "-L" + CSRC, "-lkcount_mi355", "-Wl,-rpath," + CSRC])
    return exe


def test_driver_surface_compiles_against_the_adapter(tmp_path):
    """tests/cpp/test_surface.cpp: every driver expression src/kcount/kcount_gpu.cpp uses (pass_type / PASS_TYPE, the
    member built from `{}`, get_stats().dropped, get_qf_load_factor, done_ctg_kmer_inserts, ...), inside namespace
    kcount_gpu with INTEGRATION.md's alias block, compiles and links against kcount_driver.hpp"""
    assert os.path.exists(build_surface(tmp_path))


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference checkout is not on this machine")
def test_driver_surface_is_valid_against_the_reference_headers():
    """the same text, unmodified, against the reference's own parse_and_pack.hpp / gpu_hash_table.hpp (syntax only: the
    reference's device library cannot be built here): what the adapter accepts is what the reference's host file writes"""
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-DSURFACE_REFERENCE", "-I" + REFERENCE, "-I" + os.path.join(REFERENCE, "kcount"), SURFACE])


def test_integration_md_alias_block_is_the_one_the_surface_test_compiles():
    """INTEGRATION.md's edit and the alias block of tests/cpp/test_surface.cpp are the same lines"""
    text = open(SURFACE).read()
    block = text.split("// ---- the edit INTEGRATION.md prescribes for src/kcount/kcount_gpu.cpp, verbatim ----")[1].split("// ---- end of the edit ----")[0]
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for line in block.strip().splitlines():
        line = line.strip()
        if line.startswith("#include"):
            continue
        assert line in doc, "INTEGRATION.md lacks: " + line


@pytest.mark.gpu
@pytest.mark.parametrize("k", [21, 51, 77])
def test_driver_surface_runs_like_kcount_gpu_cpp(tmp_path, k):
    """the surface program run once: three target ranks, the reference's own sequence of calls, result = the oracle's"""
    exe = build_surface(tmp_path)
    rng = np.random.default_rng(79 + k)
    reads, quals = random_reads(rng, 500, min_len=25, max_len=160, genome_len=1800)
    masked = ["".join(c.lower() if ord(x) < 33 + 20 else c for c, x in zip(r, q)) for r, q in zip(reads, quals)]
    out = subprocess.run([exe, str(k)], input="\n".join(masked) + "\n", capture_output=True, text=True, check=True).stdout
    (keys, counts, left, right), st = O.count_reads(reads, quals, k=k)
    want = sorted("%s %d %s %s" % (O.kmer_to_string(keys[i], k), counts[i], chr(left[i]), chr(right[i])) for i in range(len(counts)))
    assert out.splitlines() == want and len(want) > 50


def test_shard_bucket_ranges_tile_the_buckets(tmp_path):
    """csrc/kc_shard.hpp's ownership arithmetic on the host (tests/cpp/test_shard_ranges.cpp): every bucket has exactly one
    owner for every bucket and shard count, shard_of_bucket agrees with the ranges."""
    exe = os.path.join(str(tmp_path), "test_shard_ranges")
    subprocess.check_call([HIPCC, "-std=c++17", "-O1", "--offload-arch=gfx950", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_shard_ranges.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("bad=0"), out.stdout + out.stderr
