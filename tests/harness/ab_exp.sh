#!/bin/bash
# builds the library with each EXTRA flag set given as arguments and times it (run on the GPU box);
# AB_BIG=1 also times the large scenes (131k-triangle height field, 991k-triangle split suzanne)
cd $GRAFT_REPO_ROOT/raytracer_amd/csrc
for X in "$@"; do
  make -s clean >/dev/null; make -s EXTRA="$X" 2>/dev/null || { echo "build failed: $X"; continue; }
  echo "== EXTRA='$X'"
  for i in 1 2; do timeout -k 10 120 python $GRAFT_REPO_ROOT/tests/harness/quick_bench.py suzanne 1920 1080 16 16 0 | cut -c1-200; done
  if [ -n "$AB_BIG" ]; then
    timeout -k 10 120 python $GRAFT_REPO_ROOT/tests/harness/big_bench.py -256 4 | cut -c1-120
    timeout -k 10 120 python $GRAFT_REPO_ROOT/tests/harness/big_bench.py 5 4 | cut -c1-120
  fi
done
