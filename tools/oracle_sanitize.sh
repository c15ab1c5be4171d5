#!/bin/bash
# The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool):
# builds both math modes into a scratch directory and runs the oracle-only tests against them.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/soc_oracle_san}
mkdir -p $OUT
for mode in soc libm; do
    flag=""; [ $mode = libm ] && flag="-DSOC_ORACLE_LIBM"
    gcc -O1 -g -std=gnu11 -fPIC -shared -fopenmp -ffp-contract=off -fno-fast-math -mfma -msse4.1 \
        -fsanitize=address,undefined -fno-sanitize-recover=undefined $flag \
        -o $OUT/liborc_$mode.so $ROOT/oracle/soc_oracle.c $ROOT/oracle/a2e_oracle.c -lm
done
cd $ROOT
SOC_ORACLE_LIB_DIR=$OUT ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) \
    python -m pytest tests/test_oracle_golden.py tests/test_sca_oracle.py tests/test_maps.py tests/test_a2e.py tests/test_rng.py \
    -q -m "not gpu" -p no:cacheprovider
