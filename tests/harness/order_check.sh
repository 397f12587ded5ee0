#!/bin/bash
# first launches (no costs known yet) and later ones on the large scenes and on suzanne
cd "$(dirname "$0")/../.."
for s in "4 hf256" "4 x5s" "8 x5" "16 suzanne"; do timeout -k 10 200 python tests/harness/prof_run.py $s 2>&1 | grep -v amdgpu; done
bash tests/harness/large_now.sh 2>&1 | head -3
bash tests/harness/ab_tileorder.sh 2>&1 | grep -v amdgpu | cut -c1-150
