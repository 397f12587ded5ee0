#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of the default bench command,
# then the PMC passes on the same workload (16 frames of scene_suzanne 1080p x 16 spp).
# Results land in gpurun_out/<tag>/ and are summarised by pmc_summary.py.
set -e
TAG=${1:-profiles_r01}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kt -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_kt.log
cp $O/bench_kt/*/*_kernel_stats.csv $O/bench_kernel_stats.csv
bash $R/tests/harness/pmc.sh $TAG/pmc 16
python3 $R/tests/harness/pmc_summary.py $O/pmc > $O/pmc_summary.json
echo profiles done
