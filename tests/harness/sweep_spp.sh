#!/bin/bash
# refill threshold by samples per pixel (unfenced single-frame loop, scene_suzanne and scene_boxes 1080p); 0 = the default
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "1 0" "1 12" "1 20" "2 0" "2 12" "4 0" "4 6" "4 12" "8 0" "8 4" "8 6" "8 12" "16 0" "16 2" "16 6"; do
  set -- $cfg
  echo "spp $1 RT_REFILL=$2 :: $(RT_REFILL=$2 QB_SINGLE=1 timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 64 $1 0 2>/dev/null | grep -o 'ms/frame [0-9.]*') | $(RT_REFILL=$2 QB_SINGLE=1 timeout -k 10 120 python tests/harness/quick_bench.py boxes 1920 1080 64 $1 0 2>/dev/null | grep -o 'ms/frame [0-9.]*')"
done
done
