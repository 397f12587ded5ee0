#!/bin/bash
# bench.py (default configuration and the 991k-triangle scene) on every build of the library under raytracer_amd/csrc/_variants/
cd "$(dirname "$0")/../.."
run() {
  python bench.py --no-cpu-baseline $2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1', '$2', '%.1f Mrays/s %.3f ms/frame' % (d['value'], d['config']['ms_per_frame']))"
}
for rep in 1 2; do
  unset RT_HIP_LIBRARY
  timeout -k 10 200 bash -c "$(declare -f run); run default ''" || exit 1
  timeout -k 10 200 bash -c "$(declare -f run); run default '--config e_visible --steps 8 --warmup 2'" || exit 1
  for so in raytracer_amd/csrc/_variants/*.so; do
    export RT_HIP_LIBRARY=$PWD/$so
    timeout -k 10 200 bash -c "$(declare -f run); run $(basename $so .so) ''" || exit 1
    timeout -k 10 200 bash -c "$(declare -f run); run $(basename $so .so) '--config e_visible --steps 8 --warmup 2'" || exit 1
  done
done
