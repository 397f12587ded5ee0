#!/bin/bash
# where the large scenes stand: ms/frame and per-phase statistics
cd "$(dirname "$0")/../.."
timeout -k 10 200 python bench.py --config e_visible --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('e_visible', d['value'], d['ms_per_step'], d['roofline']['per_ray'])"
timeout -k 10 200 python bench.py --config e --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('e', d['value'], d['ms_per_step'], d['roofline']['per_ray'])"
timeout -k 10 200 python tests/harness/big_bench.py -256 4 2>&1 | grep -v amdgpu
PS_FRAMES=4 PS_BOUNCES=8 timeout -k 10 300 python tests/harness/phase_stats.py sub5x32 2>&1 | grep -v amdgpu
PS_FRAMES=4 PS_BOUNCES=8 timeout -k 10 300 python tests/harness/phase_stats.py hf256 2>&1 | grep -v amdgpu
