#!/bin/bash
# builds the library with each EXTRA flag set and times suzanne (16 spp and 1 spp), boxes, the 131k height field and the 991k mesh
cd $GRAFT_REPO_ROOT/raytracer_amd/csrc
for X in "$@"; do
  make -s clean >/dev/null; make -s EXTRA="$X" 2>/dev/null || { echo "build failed: $X"; continue; }
  echo "== EXTRA='$X'"
  for rep in 1 2; do timeout -k 10 120 python $GRAFT_REPO_ROOT/tests/harness/quick_bench.py suzanne 1920 1080 16 16 0 2>/dev/null | grep -o "ms/frame.*Mrays/s [0-9.]*" || exit 1; done
  timeout -k 10 120 python $GRAFT_REPO_ROOT/tests/harness/quick_bench.py suzanne 1920 1080 64 1 0 2>/dev/null | grep -o "ms/frame.*Mrays/s [0-9.]*" || exit 1
  timeout -k 10 120 python $GRAFT_REPO_ROOT/tests/harness/quick_bench.py boxes 1920 1080 16 16 0 2>/dev/null | grep -o "ms/frame.*Mrays/s [0-9.]*" || exit 1
  timeout -k 10 120 python $GRAFT_REPO_ROOT/tests/harness/big_bench.py -256 4 2>/dev/null | cut -c1-80 || exit 1
  timeout -k 10 120 python $GRAFT_REPO_ROOT/tests/harness/big_bench.py 5 8 2>/dev/null | cut -c1-80 || exit 1
done
make -s clean >/dev/null; make -s 2>/dev/null
