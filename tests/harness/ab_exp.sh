#!/bin/bash
# builds the library with each EXTRA flag set given as arguments and times it (run on the GPU box)
cd $GRAFT_REPO_ROOT/raytracer_amd/csrc
for X in "$@"; do
  make -s clean >/dev/null; make -s EXTRA="$X" 2>/dev/null || { echo "build failed: $X"; continue; }
  echo "== EXTRA='$X'"
  for i in 1 2; do timeout -k 10 120 python $GRAFT_REPO_ROOT/tests/harness/quick_bench.py suzanne 1920 1080 16 16 0 | cut -c1-200; done
done
