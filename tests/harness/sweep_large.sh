#!/bin/bash
# one-at-a-time sweep of the scheduling knobs on the two large scenes
cd $GRAFT_REPO_ROOT
run() { echo "$* :: $(env "$@" timeout -k 10 120 python tests/harness/big_bench.py -256 4 2>/dev/null | grep -o 'ms/frame [0-9.]*') | $(env "$@" timeout -k 10 120 python tests/harness/big_bench.py 5 8 2>/dev/null | grep -o 'ms/frame [0-9.]*')"; }
run RT_X=0
for v in 1 4 8 16 32; do run RT_SDONE=$v; done
for v in 4 8 16 24 32; do run RT_LEAF=$v; done
for v in 1 3 4 5; do run RT_SHADE=$v; done
for v in 4 8 24 48 0; do run RT_SHADEWAIT=$v; done
for v in 2 8 16; do run RT_REFILL=$v; done
