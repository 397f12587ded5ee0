#!/bin/bash
# A/B of the lane migration threshold (RT_TUNE_MIGRATE_LANES; needs a build with -DRT_MIGRATE=1)
set -e
cd "$(dirname "$0")/../.."
for m in 0 16 32; do
  RT_MIGRATE=$m timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 16 16
  RT_MIGRATE=$m RT_OVERLAP=0 RT_COALESCE=0 QB_SINGLE=1 timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 8 16
  RT_MIGRATE=$m RT_BATCH=2 timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 16 16
  timeout -k 10 200 python tests/harness/strong_scaling.py --configs c --ranks 8 --tune 13=$m
done
