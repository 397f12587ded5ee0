#!/bin/bash
# triangles per leaf and SAH triangle cost for the large scenes (the leaf loop makes full leaves cheap)
cd "$(dirname "$0")/../.."
for ml in 4 6 8; do for tc in 12 6; do
  echo "max_leaf $ml tri_cost_x10 $tc"
  RT_MAXLEAF=$ml RT_TRICOST=$tc timeout -k 10 200 python tests/harness/big_bench.py 5 8 32 2>&1 | grep -v amdgpu
  RT_MAXLEAF=$ml RT_TRICOST=$tc timeout -k 10 200 python tests/harness/big_bench.py -256 4 2>&1 | grep -v amdgpu
done; done
