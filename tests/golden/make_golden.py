#!/usr/bin/env python3
"""Writes the fixtures of tests/golden/: inputs and expected outputs, data only.

The reference's own code cannot produce them here (src/kcount/kcount_cpu.cpp and src/kmer.cpp need UPC++,
which this image lacks; the reference ships no fixtures for this path -- SURVEY.md F9/F10), so:
  * primitives.json  -- the known answers SURVEY.md section 8c recorded from the reference's src/kmer.cpp and
                        src/hash_funcs.c (copied here as data), plus values of oracle/_ref
                        (the reference's src/hash_funcs.c compiled unmodified) when that library is present;
  * hand_cases.json  -- small cases derived by hand from spec S1-S9 (each carries its derivation);
  * seeded_*.json    -- seeded random read sets with the oracle's sorted output: regression fixtures that
                        freeze today's oracle (they pin drift, not the reference).
Run from the repo root:  python tests/golden/make_golden.py
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import random_reads  # noqa: E402
from oracle import cpu_oracle as O  # noqa: E402


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
        f.write("\n")


def primitives():
    out = {
        "source": "SURVEY.md section 8c (values obtained from the reference's src/kmer.cpp + src/hash_funcs.c)",
        "pack": [
            {"k": 21, "seq": "ACGTTGCATGCATGCCGATTA", "words": ["0x1be4e4e58f000000"], "rc_words": ["0xc369393906c00000"]},
            {"k": 51, "seq": "ACGTTGCATGCATGCCGATTACGTAGCTAGCTAGCTAGCAAGGTTCCAGTC",
             "words": ["0x1be4e4e58f1b2727", "0x27242bd4b4000000"], "rc_words": ["0x87a05f9c9c9c9c6c", "0x369393906c000000"],
             "hash": "0x13b6161f7916ca8c", "minimizer_hash_m27": "0x4b08a00c5e9620ae"},
        ],
        "n_to_g": {"k": 21, "read": "ACGTTGCATGCATGCCGATNACG", "first_kmer_as": "ACGTTGCATGCATGCCGATGA",
                   "hash": "0x27eff04cccfdae09", "minimizer_hash_m15": "0x248300182c14cdb3"},
        "quick_hash": {"0": "0x7b439d0c1fd00de3", "1": "0xbea952a971ba8e83"},
        "murmur3_x64_64_of_u64_1": "0x3b35d9502fc3eead",
        "dmin_dyn": {"30": 2, "40": 3, "50": 4, "100": 9, "65535": 6553},
    }
    ref = os.path.join(ROOT, "oracle", "_ref", "libref_hash_funcs.so")
    if os.path.exists(ref):
        L = C.CDLL(ref)
        L.MurmurHash3_x64_64.restype = C.c_uint64
        L.MurmurHash3_x64_64.argtypes = [C.c_void_p, C.c_uint32]
        L.quick_hash.restype = C.c_uint64
        L.quick_hash.argtypes = [C.c_uint64]
        rng = np.random.default_rng(2024)
        vec = []
        for n in (0, 1, 7, 8, 9, 15, 16, 17, 24, 32, 33):
            buf = rng.integers(0, 256, size=max(n, 1), dtype=np.uint8)
            vec.append({"bytes": bytes(buf[:n]).hex(), "murmur3_x64_64": hex(L.MurmurHash3_x64_64(buf.ctypes.data, n))})
        out["ref_hash_funcs"] = {"source": "oracle/_ref/libref_hash_funcs.so = reference src/hash_funcs.c compiled unmodified",
                                 "murmur": vec,
                                 "quick_hash": {str(v): hex(L.quick_hash(v)) for v in (2, 12345, 2**40 + 7, 2**64 - 1)}}
    dump("primitives.json", out)


def hand_cases():
    cases = [
        {"name": "clean repeat with a self-overlap", "k": 5, "reads": ["ACGTACGGA"] * 2, "quals": None,
         "why": "i=1 CGTAC(A,G); i=2 GTACG->rc CGTAC(C,G); i=3 TACGG->rc CCGTA(T,C). CGTAC: left A:2 C:2 -> F, purged",
         "expect": ["CCGTA 2 T C"]},
        {"name": "reverse strand duplicates swap and complement the extensions", "k": 5, "reads": ["AACCGTAG", "CTACGGTT"], "quals": None,
         "why": "second read is the reverse complement of the first", "expect": ["ACCGT 2 A A", "CCGTA 2 A G"]},
        {"name": "fork", "k": 5, "reads": ["AACCGTAG"] * 2 + ["AACCGTAC"] * 2, "quals": None,
         "why": "CCGTA is followed by G twice and C twice -> F", "expect": ["ACCGT 4 A A"]},
        {"name": "low-quality neighbour", "k": 5, "reads": ["AACCGTAG"] * 2, "quals": ["IIIIIIII", "#IIIIIII"],
         "why": "left neighbour of ACCGT is low quality once: one vote < dmin 2 -> X", "expect": ["CCGTA 2 A G"]},
        {"name": "N inside a k-mer counts as G", "k": 5, "reads": ["AACCNTAG", "AACCGTAG"], "quals": None,
         "why": "src/kmer.cpp:173,191-192", "expect": ["ACCGT 2 A A", "CCGTA 2 A G"]},
        {"name": "palindrome keeps forward extensions", "k": 4, "reads": ["TACGTC"] * 2, "quals": None,
         "why": "strict < at kcount_cpu.cpp:328", "expect": ["ACGT 2 T C"]},
        {"name": "single occurrences are purged", "k": 5, "reads": ["AACCGTAG"], "quals": None, "why": "count < 2", "expect": []},
        {"name": "reads shorter than k+2 contribute nothing", "k": 5, "reads": ["ACGTAC", "ACGTA", "ACG", ""], "quals": None,
         "why": "S1", "expect": []},
    ]
    dump("hand_cases.json", cases)


def seeded():
    for k, seed, n in ((21, 11, 2500), (33, 12, 1500), (51, 13, 1500), (77, 14, 1200)):
        rng = np.random.default_rng(seed)
        reads, quals = random_reads(rng, n, min_len=max(4, k - 4), max_len=k + 110, genome_len=4000)
        (keys, counts, left, right), st = O.count_reads(reads, quals, k=k, nranks=3, nthreads=2)
        lines = ["%s %d %s %s" % (O.kmer_to_string(keys[i], k), counts[i], chr(left[i]), chr(right[i])) for i in range(len(counts))]
        dump("seeded_k%d.json" % k, {
            "k": k, "generator": {"fn": "tests/helpers.py::random_reads", "seed": seed, "nreads": n,
                                  "kwargs": {"min_len": max(4, k - 4), "max_len": k + 110, "genome_len": 4000}},
            "input_sha256": hashlib.sha256(("\n".join(reads) + "|" + "\n".join(quals)).encode()).hexdigest(),
            "stats": {s: st[s] for s in ("raw_kmers", "kmers_inserted", "unique", "purged", "total_kmers", "sum_counts")},
            "num_lines": len(lines),
            "lines_sha256": hashlib.sha256("\n".join(lines).encode()).hexdigest(),
            "first_lines": lines[:20], "last_lines": lines[-5:],
        })


if __name__ == "__main__":
    primitives()
    hand_cases()
    seeded()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".json")))
