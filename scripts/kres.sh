#!/bin/bash
# per-kernel resources of the hot kernels of an experiment build: scripts/kres.sh [extra hipcc flags]
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$R/mhm2_kmer_analysis_v2_amd/csrc
TMP=$SRC/.fast_res_$$.hip
sed -E -e '/^\s*(case [23]|default):.*<[234][,>]/d' -e '/case 1:.*<1, FMT_(READS_UQ|PACKED|SEQBLOCK)>/d' $SRC/kc_api.hip > $TMP
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-return-type -ffp-contract=off -Rpass-analysis=kernel-resource-usage "$@" -c -o /dev/null $TMP 2>&1 | python3 -c "
import sys,re
cur=None
want=('l1_reads_kernelILi1ELi0ELb1ELb0ELi21','l1_reads16_kernelILi0ELb0','l2_split_kernelILi1ELb1ELb1ELb0','l2_rec6_kernelILb0ELb0','bin16_kernelILi0ELi21','l1_wire6_kernel','count_kernelILi1ELb0ELb1','l1_slots_kernelILi0ELb0ELi21','l2_slots_kernelILb0')
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); continue
    for key in ('VGPRs:','ScratchSize','VGPRs Spill','SGPRs Spill'):
        if key in l and cur and any(x in cur for x in want):
            print(cur[9:46], l.split('remark:')[1].strip().replace(' [-Rpass-analysis=kernel-resource-usage]',''))
"
rm -f $TMP
