#!/bin/bash
# what-if timings on the GPU box: occupancy sweep with the committed build, then builds that drop
# work (not parity-clean; timing only) to see how much of the frame time each part accounts for
cd $GRAFT_REPO_ROOT
for B in 3 4 5 6; do RT_BPC=$B timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 16 16 0 2>/dev/null | cut -c1-110; done
cd raytracer_amd/csrc
for X in "$@"; do
  make -s clean >/dev/null; make -s EXTRA="$X" 2>/dev/null || { echo "build failed: $X"; continue; }
  echo "== EXTRA='$X'"
  timeout -k 10 120 python $GRAFT_REPO_ROOT/tests/harness/quick_bench.py suzanne 1920 1080 16 16 0 2>/dev/null | cut -c1-110
done
make -s clean >/dev/null; make -s 2>/dev/null
