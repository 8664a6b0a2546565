#!/bin/bash
# ISA of one kernel of an experiment build: scripts/asm_kernel.sh MANGLED_NAME_PREFIX OUT.s [extra hipcc flags]
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$R/mhm2_kmer_analysis_v2_amd/csrc
NAME=$1; OUT=$2; shift 2
TMP=$SRC/.fast_asm_$$.hip
sed -E -e '/^\s*(case [23]|default):.*<[234][,>]/d' -e '/case 1:.*<1, FMT_(READS_UQ|PACKED|SEQBLOCK)>/d' $SRC/kc_api.hip > $TMP
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-return-type -ffp-contract=off -S --cuda-device-only "$@" -o /tmp/kc_all_$$.s $TMP 2>/dev/null
rm -f $TMP
awk -v n="$NAME" 'index($0,n)==1 && /:/{p=1} p{print} p&&/s_endpgm/{exit}' /tmp/kc_all_$$.s > $OUT
rm -f /tmp/kc_all_$$.s
wc -l $OUT
