#!/bin/bash
# one bench configuration ("$@" = bench.py arguments) on the in-tree library and on every build under raytracer_amd/csrc/_variants/
cd "$(dirname "$0")/../.."
one() { python bench.py --no-cpu-baseline "${@:2}" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1', '%.1f Mrays/s %.4f ms/frame' % (d['value'], d['config']['ms_per_frame']))"; }
for rep in 1 2 3; do
  unset RT_HIP_LIBRARY; one in-tree "$@" || exit 1
  for so in raytracer_amd/csrc/_variants/*.so; do
    RT_HIP_LIBRARY=$PWD/$so one $(basename $so .so) "$@" || exit 1
  done
done
