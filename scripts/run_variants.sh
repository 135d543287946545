#!/bin/bash
# usage: scripts/run_variants.sh NAME...  -- bench_group.py against the product library and each built variant
V=pgl_amd/csrc/variants
F=${FILTER:-'LIB|float32  d=(48|64|128) |float16  d=128 |float64  d=32 '}
timeout 60 python scripts/bench_group.py 2>&1 | grep -E "$F"
for n in "$@"; do PGLAMD_LIB=$PWD/$V/libpglamd_$n.so timeout 60 python scripts/bench_group.py 2>&1 | grep -E "$F"; done
