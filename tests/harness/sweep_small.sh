#!/bin/bash
# one-at-a-time sweep of the scheduling knobs on scene_suzanne and scene_boxes (1080p, 16 frames x 16 spp)
cd $GRAFT_REPO_ROOT
run() { echo "$* :: $(env "$@" timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 16 16 0 2>/dev/null | grep -o 'ms/frame [0-9.]*') | $(env "$@" timeout -k 10 120 python tests/harness/quick_bench.py boxes 1920 1080 16 16 0 2>/dev/null | grep -o 'ms/frame [0-9.]*')"; }
run RT_X=0
run RT_X=1
for v in 1 4 8; do run RT_SDONE=$v; done
for v in 6 8 10 16 20; do run RT_LEAF=$v; done
for v in 1 3 4; do run RT_SHADE=$v; done
for v in 6 8 16 24 0; do run RT_SHADEWAIT=$v; done
for v in 2 6 8; do run RT_REFILL=$v; done
