#!/bin/bash
# the figures quoted for the configurations other than the default bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out
for cfg in a b d e e_visible; do
  timeout -k 10 300 python bench.py --config $cfg --steps 4 --no-cpu-baseline > $O/r3_final_bench_$cfg.json 2> $O/r3_final_bench_$cfg.err || { echo "bench $cfg failed"; exit 1; }
  python - $O/r3_final_bench_$cfg.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(d["config"]["workload"][:70], "| ms/frame %.3f | Grays/s %.2f" % (d["config"]["ms_per_frame"], d["value"] / 1e3))
PY
done
timeout -k 10 120 python tests/harness/big_bench.py -256 4 | cut -c1-90 || exit 1
QB_SINGLE=1 timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 32 16 0 | grep -o "ms/frame [0-9.]*" || exit 1
QB_SINGLE=1 timeout -k 10 120 python tests/harness/quick_bench.py suzanne 1920 1080 128 1 0 | grep -o "ms/frame [0-9.]*" || exit 1
timeout -k 10 400 python bench.py > $O/r3_final_bench_n1.json 2> $O/r3_final_bench_n1.err || { echo "default bench failed"; exit 1; }
cut -c1-160 $O/r3_final_bench_n1.json
