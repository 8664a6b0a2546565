"""mhm2_kmer_analysis_v2_amd -- MI355X-native k-mer analysis (kcount) stage.

The product is csrc/libkcount_mi355.so (hand-written HIP for gfx950 behind the C
ABI of include/kcount_mi355.h).  This package is the thin Python host side used
by the tests and bench.py: a ctypes binding plus a driver that mirrors the
reference's analyze_kmers flow (src/kcount/kcount.cpp:142-161).  There is no CPU
fallback: without the built library or without a GPU every call fails loudly.
"""
from ._lib import KcError, lib, lib_path  # noqa: F401
from .kcount import KmerCounter, analyze_kmers, fastq_to_packed, synth_params, synth_reads_host  # noqa: F401
