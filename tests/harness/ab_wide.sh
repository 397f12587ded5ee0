#!/bin/bash
# the in-tree library against raytracer_amd/csrc/_variants/$1.so on the other workloads: configs a, b, d, e, the height field, 1 spp
cd "$(dirname "$0")/../.."
V=$PWD/raytracer_amd/csrc/_variants/$1.so
one() { python bench.py --no-cpu-baseline $1 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$2', '$1', '%.1f Mrays/s %.3f ms/frame' % (d['value'], d['config']['ms_per_frame']))"; }
for rep in 1 2; do
  for cfg in "--config a --steps 16" "--config b --steps 16" "--config d --steps 4" "--config e --steps 8"; do
    unset RT_HIP_LIBRARY; timeout -k 10 200 bash -c "$(declare -f one); one '$cfg' in-tree" || exit 1
    export RT_HIP_LIBRARY=$V; timeout -k 10 200 bash -c "$(declare -f one); one '$cfg' $1" || exit 1
  done
  unset RT_HIP_LIBRARY; echo in-tree; timeout -k 10 120 python tests/harness/big_bench.py -256 4 2>&1 | grep -o "ms/frame [0-9.]*"; QB_SINGLE=1 timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 128 1 0 | grep -o "ms/frame [0-9.]*"; QB_SINGLE=1 timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 32 16 0 | grep -o "ms/frame [0-9.]*"
  export RT_HIP_LIBRARY=$V; echo $1; timeout -k 10 120 python tests/harness/big_bench.py -256 4 2>&1 | grep -o "ms/frame [0-9.]*"; QB_SINGLE=1 timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 128 1 0 | grep -o "ms/frame [0-9.]*"; QB_SINGLE=1 timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 32 16 0 | grep -o "ms/frame [0-9.]*"
done
