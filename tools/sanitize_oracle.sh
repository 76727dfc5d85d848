#!/bin/bash
# Build the CPU oracle with AddressSanitizer + UBSan and run the oracle test-suite against it (CPU only; GPU sanitizers are
# not available on the pool).  Restores the normal build afterwards.
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
make -C oracle -B CXXFLAGS="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread -fsanitize=address,undefined -fno-sanitize-recover=undefined" > /dev/null || exit 1
LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_oracle_stages.py tests/test_oracle_halfconv.py tests/test_golden.py -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -5
rc=${PIPESTATUS[0]}
make -C oracle -B > /dev/null
exit $rc
