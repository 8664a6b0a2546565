"""Shared test helpers: random read sets and result comparison."""
import numpy as np


def random_reads(rng, nreads, min_len=10, max_len=120, genome_len=600, err=0.02, n_rate=0.01, lowq_rate=0.05,
                 qual_offset=33):
    """Reads sampled from a small random genome (both strands) so that k-mers repeat."""
    genome = "".join(rng.choice(list("ACGT"), size=genome_len))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    reads, quals = [], []
    for _ in range(nreads):
        ln = int(rng.integers(min_len, max_len + 1))
        ln = min(ln, genome_len)
        st = int(rng.integers(0, genome_len - ln + 1))
        s = list(genome[st:st + ln])
        if rng.random() < 0.5:
            s = [comp[c] for c in reversed(s)]
        q = []
        for i in range(ln):
            r = rng.random()
            if r < err:
                s[i] = "ACGT"[int(rng.integers(0, 4))]
            elif r < err + n_rate:
                s[i] = "N"
            q.append(chr(qual_offset + (2 if rng.random() < lowq_rate else 40)))
        reads.append("".join(s))
        quals.append("".join(q))
    return reads, quals


def results_to_tuples(keys, counts, left, right):
    """numpy results -> sorted list of (key words tuple, count, L, R)."""
    out = [(tuple(int(x) for x in keys[i]), int(counts[i]), chr(left[i]), chr(right[i])) for i in range(len(counts))]
    out.sort()
    return out
