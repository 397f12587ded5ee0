#!/bin/bash
# shade threshold x shade wait on the three large scenes (height field, 991k triangles unseen and seen)
cd $GRAFT_REPO_ROOT
run() { echo "$* :: $(env "$@" timeout -k 10 120 python tests/harness/big_bench.py -256 4 2>/dev/null | grep -o 'ms/frame [0-9.]*') | $(env "$@" timeout -k 10 120 python tests/harness/big_bench.py 5 8 2>/dev/null | grep -o 'ms/frame [0-9.]*') | $(env "$@" timeout -k 10 120 python tests/harness/big_bench.py 5 8 32 2>/dev/null | grep -o 'ms/frame [0-9.]*')"; }
for s in 2 3 4; do for w in 8 12 16; do run RT_SHADE=$s RT_SHADEWAIT=$w; done; done
run RT_SHADE=3 RT_SHADEWAIT=12 RT_LEAF=8
run RT_SHADE=3 RT_SHADEWAIT=12 RT_LEAF=16
