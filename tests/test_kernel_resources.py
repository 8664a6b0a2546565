"""The hot kernels' register allocation, read from the code object inside the SHIPPED library (no recompilation): the
instantiations the benchmark configuration runs must not spill vector registers in their loops -- a reload from scratch is
a vector-memory load, and its wait is a wait for every prefetch load and copy-out store the wave has in flight
(csrc/kc_bucketed.hpp, fresh_tid)."""
import os
import re
import subprocess
import tempfile

import pytest

import mhm2_kmer_analysis_v2_amd as pkg

LLVM = "/opt/rocm/lib/llvm/bin"
needs_llvm = pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "clang-offload-bundler")), reason="ROCm's LLVM tools not found")


def kernel_metadata():
    so = pkg._lib.library_path() if hasattr(pkg._lib, "library_path") else os.path.join(os.path.dirname(pkg.__file__), "csrc", "libkcount_mi355.so")
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fat])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    out, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s+\.(name|private_segment_fixed_size|vgpr_count|vgpr_spill_count|sgpr_spill_count):\s+(\S+)", line)
        if not m:
            continue
        if m.group(1) == "name":
            cur = out.setdefault(m.group(2), {})
        elif cur is not None:
            cur[m.group(1)] = int(m.group(2))
    return out


@needs_llvm
def test_hot_kernels_do_not_spill_vector_registers():
    md = kernel_metadata()
    assert len(md) > 50

    def one(prefix):
        hits = [(n, v) for n, v in md.items() if n.startswith(prefix)]
        assert len(hits) == 1, (prefix, [n for n, _ in hits])
        return hits[0][1]

    # level 1 of the benchmark configuration: sixteen k-mers per thread and round, ASCII reads, one shard, k = 21
    l1 = one("_ZN2kc20kc_l1_reads16_kernelILi0ELb0ELi21EE")
    assert l1["vgpr_spill_count"] == 0 and l1["private_segment_fixed_size"] == 0 and l1["vgpr_count"] <= 128, l1
    # the count kernel of compact records: two workgroups per CU need at most 64 registers
    cnt = one("_ZN2kc15kc_count_kernelILi1ELb0ELb1EE")
    assert cnt["vgpr_spill_count"] == 0 and cnt["private_segment_fixed_size"] == 0 and cnt["vgpr_count"] <= 64, cnt
    # level 2 of the benchmark configuration (six-byte level-1 records, k = 21): nothing spilled
    l26 = one("_ZN2kc17kc_l2_rec6_kernelILb0ELb0EE")
    assert l26["vgpr_spill_count"] == 0 and l26["private_segment_fixed_size"] == 0 and l26["vgpr_count"] <= 128, l26
    # the records flow's two kernels of the six-byte wire record (csrc/kc_wire6.hpp): nothing spilled
    for name in ("_ZN2kc15kc_bin16_kernelILi0ELi21EE", "_ZN2kc18kc_l1_wire6_kernelE"):
        w6 = one(name)
        assert w6["vgpr_spill_count"] == 0 and w6["private_segment_fixed_size"] == 0 and w6["vgpr_count"] <= 128, (name, w6)
    # level 2 of the other short-form compact records: three registers are spilled in the per-BUCKET prologue and epilogue (a pair of
    # zeros and the lane id, four times per workgroup and step), none inside the round loop
    l2 = one("_ZN2kc18kc_l2_split_kernelILi1ELb1ELb1ELb0ELb0EE")
    assert l2["vgpr_spill_count"] <= 3 and l2["private_segment_fixed_size"] <= 16 and l2["vgpr_count"] <= 128, l2
