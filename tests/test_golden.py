"""The committed fixtures of tests/golden/ (made by tests/golden/make_golden.py) against the oracle (CPU) and the
HIP path (GPU)."""
import hashlib
import json
import os

import numpy as np
import pytest

from helpers import random_reads
from oracle import cpu_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def lines_of(res, k):
    keys, counts, left, right = res
    return ["%s %d %s %s" % (O.kmer_to_string(keys[i], k), counts[i], chr(left[i]), chr(right[i])) for i in range(len(counts))]


def seeded_input(g):
    gen = g["generator"]
    rng = np.random.default_rng(gen["seed"])
    reads, quals = random_reads(rng, gen["nreads"], **gen["kwargs"])
    assert hashlib.sha256(("\n".join(reads) + "|" + "\n".join(quals)).encode()).hexdigest() == g["input_sha256"], \
        "the seeded generator no longer reproduces the fixture's input"
    return reads, quals


def test_primitive_known_answers():
    p = load("primitives.json")
    for e in p["pack"]:
        w = O.pack_kmer(e["seq"])
        assert [hex(int(x)) for x in w] == e["words"]
        assert [hex(int(x)) for x in O.revcomp(w, e["k"])] == e["rc_words"]
        if "hash" in e:
            assert hex(O.kmer_hash(w)) == e["hash"]
            assert hex(O.minimizer_hash(w, e["k"], 27)) == e["minimizer_hash_m27"]
    n = p["n_to_g"]
    km = O.get_kmers(n["read"], n["k"])[0]
    assert (km == O.pack_kmer(n["first_kmer_as"])).all()
    assert hex(O.kmer_hash(km)) == n["hash"] and hex(O.minimizer_hash(km, n["k"], 15)) == n["minimizer_hash_m15"]
    for v, h in p["quick_hash"].items():
        assert hex(O.lib().orc_quick_hash(int(v))) == h
    one = np.array([1], dtype=np.uint64)
    assert hex(O.lib().orc_murmur3_x64_64(one.ctypes.data, 8)) == p["murmur3_x64_64_of_u64_1"]
    for c, d in p["dmin_dyn"].items():
        c = int(c)
        assert O.get_ext([d, 0, 0, 0], c) == "A" and O.get_ext([d - 1, 0, 0, 0], c) == "X"
    if "ref_hash_funcs" in p:  # values of the reference's own hash_funcs.c
        for e in p["ref_hash_funcs"]["murmur"]:
            buf = np.frombuffer(bytes.fromhex(e["bytes"]) or b"\0", dtype=np.uint8).copy()
            assert hex(O.lib().orc_murmur3_x64_64(buf.ctypes.data, len(e["bytes"]) // 2)) == e["murmur3_x64_64"]
        for v, h in p["ref_hash_funcs"]["quick_hash"].items():
            assert hex(O.lib().orc_quick_hash(int(v))) == h


def test_hand_cases_oracle():
    for c in load("hand_cases.json"):
        res, _ = O.count_reads(c["reads"], c["quals"], k=c["k"])
        assert lines_of(res, c["k"]) == c["expect"], c["name"]


@pytest.mark.parametrize("name", ["seeded_k21.json", "seeded_k33.json", "seeded_k51.json", "seeded_k77.json"])
def test_seeded_fixture_oracle(name):
    g = load(name)
    reads, quals = seeded_input(g)
    res, st = O.count_reads(reads, quals, k=g["k"], nranks=5, nthreads=3)
    lines = lines_of(res, g["k"])
    assert len(lines) == g["num_lines"] and lines[:20] == g["first_lines"] and lines[-5:] == g["last_lines"]
    assert hashlib.sha256("\n".join(lines).encode()).hexdigest() == g["lines_sha256"]
    for s, v in g["stats"].items():
        assert st[s] == v, s


@pytest.mark.gpu
def test_hand_cases_gpu():
    import mhm2_kmer_analysis_v2_amd as pkg
    for c in load("hand_cases.json"):
        for tuning in (None, dict(mode=1)):
            with pkg.KmerCounter(c["k"], tuning=tuning) as kc:
                kc.submit_reads(*O.reads_to_arrays(c["reads"], c["quals"]))
                assert kc.dump_lines() == c["expect"], c["name"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["seeded_k21.json", "seeded_k33.json", "seeded_k51.json", "seeded_k77.json"])
def test_seeded_fixture_gpu(name):
    import mhm2_kmer_analysis_v2_amd as pkg
    g = load(name)
    reads, quals = seeded_input(g)
    for tuning in (None, dict(writers=3, p1=4, p2=8, slots=256), dict(mode=1)):
        with pkg.KmerCounter(g["k"], tuning=tuning) as kc:
            kc.submit_reads(*O.reads_to_arrays(reads, quals))
            lines = kc.dump_lines()
            st = kc.stats()
        assert hashlib.sha256("\n".join(lines).encode()).hexdigest() == g["lines_sha256"]
        assert st["num_unique"] == g["stats"]["unique"] and st["sum_counts"] == g["stats"]["sum_counts"]
