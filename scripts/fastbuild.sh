#!/bin/bash
# Experiment builds of the library for A/B runs of the default bench only: the instantiations for several-word k-mers and
# for the input formats the bench does not use are cut out of a copy of kc_api.hip, so that a build takes seconds instead
# of most of a minute.  NOT the product build (make -C mhm2_kmer_analysis_v2_amd/csrc).
#   scripts/fastbuild.sh OUT.so [extra hipcc flags...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
SRC=$R/mhm2_kmer_analysis_v2_amd/csrc
TMP=$SRC/.fast_$$.hip
sed -E -e '/^\s*(case [23]|default):.*<[234][,>]/d' -e '/case 1:.*<1, FMT_(READS_UQ|PACKED|SEQBLOCK)>/d' $SRC/kc_api.hip > $TMP
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-return-type -ffp-contract=off -shared "$@" -o $OUT $TMP
rm -f $TMP
