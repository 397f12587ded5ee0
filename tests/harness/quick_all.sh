#!/bin/bash
# the timing set of ab_exp4.sh with the library as built
cd $GRAFT_REPO_ROOT
for rep in 1 2; do timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 16 16 0 2>/dev/null | grep -o "ms/frame.*Mrays/s [0-9.]*" || exit 1; done
timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 64 1 0 2>/dev/null | grep -o "ms/frame.*Mrays/s [0-9.]*" || exit 1
timeout -k 10 120 python tests/harness/quick_bench.py boxes 1920 1080 16 16 0 2>/dev/null | grep -o "ms/frame.*Mrays/s [0-9.]*" || exit 1
timeout -k 10 120 python tests/harness/quick_bench.py spheres 640 360 64 16 0 2>/dev/null | grep -o "ms/frame.*Mrays/s [0-9.]*" || exit 1
timeout -k 10 120 python tests/harness/big_bench.py -256 4 2>/dev/null | cut -c1-80 || exit 1
timeout -k 10 120 python tests/harness/big_bench.py 5 8 2>/dev/null | cut -c1-80 || exit 1
