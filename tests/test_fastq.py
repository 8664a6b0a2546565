"""FASTQ -> PackedReads front end (N4 of SURVEY.md 8f; src/fastq.cpp:1028-1140, src/packed_reads.cpp:99-126): a host
function of the C ABI.  CPU tests pin the format; the GPU test feeds its output to kc_submit_packed_reads and compares
with the oracle on the same reads."""
import numpy as np
import pytest

import mhm2_kmer_analysis_v2_amd as pkg
from helpers import random_reads
from oracle import cpu_oracle as O


def fastq(reads, quals, crlf=False, names=None):
    nl = "\r\n" if crlf else "\n"
    return "".join("@%s%s%s%s+%s%s%s" % ((names[i] if names else "read%d extra words" % i), nl, r, nl, nl, q, nl)
                   for i, (r, q) in enumerate(zip(reads, quals)))


def test_packed_bytes_follow_packed_read():
    reads = ["ACGTNacgtn", "RYKMSWBDHVU", "T"]
    quals = ["I#5!~IIII+", "IIIIIIIIIII", "@"]
    packed, offs = pkg.fastq_to_packed(fastq(reads, quals))
    assert offs.tolist() == [0, 10, 21, 22]
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    want = [code.get(c.upper(), 4) | (min(ord(q) - 33, 31) << 3) for r, qq in zip(reads, quals) for c, q in zip(r, qq)]
    assert packed.tolist() == want
    # CR LF line ends and trailing blanks are stripped like rtrim does (fastq.cpp:1098-1100); a last line without \n is fine
    p2, o2 = pkg.fastq_to_packed(fastq(reads, quals, crlf=True).rstrip("\r\n"))
    assert p2.tolist() == want and o2.tolist() == offs.tolist()
    # quality offset 64
    p3, _ = pkg.fastq_to_packed("@x\nAC\n+\nhB\n", qual_offset=64)
    assert p3.tolist() == [0 | (31 << 3), 1 | (2 << 3)]
    assert pkg.fastq_to_packed("")[1].tolist() == [0]


@pytest.mark.parametrize("text,status", [
    ("@x\nACGT\n+\nIII\n", -1),        # sequence and qualities differ in length (fastq.cpp:1118-1121)
    ("x\nACGT\n+\nIIII\n", -1),        # no '@' (fastq.cpp:1101)
    ("@x\nACGT\n-\nIIII\n", -1),       # no '+' (fastq.cpp:1102)
    ("@x\nACGT\n+\n", -1),             # the text ends inside a record
    ("@x\nACZT\n+\nIIII\n", -7),       # a character PackedRead DIEs on (packed_reads.cpp:121-123)
])
def test_malformed_fastq_is_an_error(text, status):
    with pytest.raises(pkg.KcError) as e:
        pkg.fastq_to_packed(text)
    assert e.value.status == status


@pytest.mark.gpu
@pytest.mark.parametrize("k", [21, 51])
def test_fastq_to_kmers_matches_oracle(k):
    rng = np.random.default_rng(300 + k)
    reads, quals = random_reads(rng, 3000, min_len=k - 3, max_len=k + 120, genome_len=5000)
    packed, offs = pkg.fastq_to_packed(fastq(reads, quals, crlf=(k == 51)))
    (keys, counts, left, right), st = O.count_reads(reads, quals, k=k, nranks=3, nthreads=3)
    with pkg.KmerCounter(k) as kc:
        kc.submit_packed_reads(packed, offs)
        got = kc.sorted_results()
        gst = kc.stats()
    for g, w in zip(got, (keys, counts, left, right)):
        assert g.shape == w.shape and (g == w).all()
    assert gst["num_reads"] == len(reads) and gst["raw_kmers"] == st["raw_kmers"]
