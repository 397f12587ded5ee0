#!/bin/bash
# builds the library with different occupancy targets and times each (run on the GPU box)
set -e
cd $GRAFT_REPO_ROOT/raytracer_amd/csrc
for W in 1 5 6 8; do
  make -s clean >/dev/null; make -s EXTRA="-DRT_MINW=$W" 2>/dev/null
  echo "== RT_MINW=$W"
  timeout -k 10 120 python $GRAFT_REPO_ROOT/tests/harness/quick_bench.py suzanne 1920 1080 16 16 0 | cut -c1-100
  RT_SHADE=2 timeout -k 10 120 python $GRAFT_REPO_ROOT/tests/harness/quick_bench.py suzanne 1920 1080 16 16 0 | cut -c1-100
done
