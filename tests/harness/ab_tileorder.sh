#!/bin/bash
# A/B of the tile hand-out order (RT_TUNE_TILE_ORDER): long launches, single fused frames, short launches, 1 spp
set -e
cd "$(dirname "$0")/../.."
for m in 0 1; do
  RT_TILEORDER=$m timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 16 16
  RT_TILEORDER=$m RT_OVERLAP=0 RT_COALESCE=0 QB_SINGLE=1 timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 8 16
  RT_TILEORDER=$m RT_COALESCE=0 QB_SINGLE=1 timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 16 16
  RT_TILEORDER=$m RT_BATCH=2 timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 16 16
  RT_TILEORDER=$m timeout -k 10 120 python tests/harness/quick_bench.py boxes 1920 1080 16 16
  RT_TILEORDER=$m timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 64 1
  RT_TILEORDER=$m RT_COALESCE=0 QB_SINGLE=1 timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 64 1
  timeout -k 10 200 python tests/harness/strong_scaling.py --configs c --ranks 8 --tune 14=$m
done
