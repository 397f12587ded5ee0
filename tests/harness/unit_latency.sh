#!/bin/bash
# how long a launch of very few pixels takes: the serial latency of the longest 16-sample pixel on an empty GPU
cd "$(dirname "$0")/../.."
for wh in "8 8" "16 16" "64 36" "128 72" "256 144"; do
  RT_OVERLAP=0 RT_COALESCE=0 QB_SINGLE=1 timeout -k 10 120 python tests/harness/quick_bench.py suzanne $wh 8 16
done
