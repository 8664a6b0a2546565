"""Second, independent restatement of spec S1-S9 in plain
Python strings and dicts.  Deliberately naive: no bit packing, no hashing, no
supermers, no ranks.  Used only to cross-check oracle/kcount_oracle.c on small
inputs -- two restatements written differently must agree.

Reference semantics restated (paths relative to /root/reference):
  src/kcount/kcount.cpp:78-86, src/kcount/kcount_cpu.cpp:308-355 (occurrences,
  extensions), :135-182 (vote), :557-573 (purge), src/kmer.cpp:173,191-192
  (N counts as G inside a k-mer), src/utils.cpp:132-159 (complement).
"""

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N", "0": "0"}
QUAL_CUTOFF = 20
CAP = 65535


def revcomp_str(s):
    return "".join(COMP[c] for c in reversed(s))


def count_kmers(reads, quals, k, qual_offset=33, dmin_thres=2):
    """reads/quals: lists of equal-length strings.  Returns
    (results, table): results = sorted list of (kmer_str, count, L, R);
    table = {kmer_str: [count, [lA,lC,lG,lT], [rA,rC,rG,rT]]} before the purge."""
    table = {}
    for seq, q in zip(reads, quals):
        seq = seq.upper()
        n = len(seq)
        if n < k:
            continue
        hq = [ord(q[i]) >= qual_offset + QUAL_CUTOFF for i in range(n)]
        kseq = seq.replace("N", "G")  # inside a k-mer an N is packed as G
        for i in range(1, n - k):
            kmer = kseq[i:i + k]
            left = seq[i - 1] if hq[i - 1] else "0"
            right = seq[i + k] if hq[i + k] else "0"
            rc = revcomp_str(kmer)
            if rc < kmer:  # ACGT order == 2-bit code order
                kmer = rc
                left, right = COMP[right], COMP[left]
            e = table.setdefault(kmer, [0, [0, 0, 0, 0], [0, 0, 0, 0]])
            e[0] = min(e[0] + 1, CAP)
            if left in "ACGT":
                e[1]["ACGT".index(left)] = min(e[1]["ACGT".index(left)] + 1, CAP)
            if right in "ACGT":
                e[2]["ACGT".index(right)] = min(e[2]["ACGT".index(right)] + 1, CAP)
    results = []
    for kmer, (count, lc, rc_) in table.items():
        if count < 2:
            continue
        l = vote(lc, count, dmin_thres)
        r = vote(rc_, count, dmin_thres)
        if l in "XF" or r in "XF":
            continue
        results.append((kmer, count, l, r))
    results.sort()
    return results, table


def vote(c4, count, dmin_thres=2):
    pairs = sorted(zip("ACGT", c4), key=lambda p: (p[1], p[0]), reverse=True)
    dmin_dyn = max(int((1.0 - 0.9) * count), dmin_thres)
    if pairs[0][1] < dmin_dyn:
        return "X"
    if pairs[1][1] >= dmin_dyn:
        return "F"
    return pairs[0][0]
