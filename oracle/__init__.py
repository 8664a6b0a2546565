"""CPU oracle for the kcount hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Nothing under mhm2_kmer_analysis_v2_amd/ imports it.
"""
