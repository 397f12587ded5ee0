#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of the default bench command, the PMC passes on
# the same workload (16 frames of scene_suzanne 1080p x 16 spp) and on the large scenes, and the
# counter calibration on the FMA microbenchmark.  Results land in gpurun_out/<tag>/.
TAG=${1:-profiles_r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kt -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_kt.log || { echo "bench under rocprof failed"; exit 1; }
cp $O/bench_kt/*/*_kernel_stats.csv $O/bench_kernel_stats.csv
bash $R/tests/harness/pmc.sh $TAG/pmc_c 16 suzanne || exit 1
bash $R/tests/harness/pmc.sh $TAG/pmc_hf256 4 hf256 || exit 1
bash $R/tests/harness/pmc.sh $TAG/pmc_x5 8 x5 || exit 1
bash $R/tests/harness/pmc.sh $TAG/pmc_x5s 4 x5s || exit 1
# calibration: one (kind, mask, waves per SIMD) per run so that the counters belong to one dispatch shape
for K in "fma 0 8" "fma 1 8" "fma 3 8" "fma 2 8" "pk 0 8" "fma 0 2"; do
  set -- $K
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU2 GRBM_GUI_ACTIVE --output-format csv -d $O/calib_$1_$2_$3 -- $R/tests/_build/valu_calib 20000 $1 $2 $3 > $O/calib_$1_$2_$3.log 2>&1 || { echo "calibration pass failed: $K"; exit 1; }
done
# vector-memory calibration: clocks per 16-byte load instruction under execution masks, then the L1 counters on the same runs
timeout -k 10 120 $R/tests/_build/valu_calib l 4000 > $O/vmem_calibration.txt 2>&1 || { echo "load calibration failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_TOTAL_WAVEFRONTS_sum TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $O/vmem_pmc -- $R/tests/_build/valu_calib l 4000 > $O/vmem_pmc.log 2>&1 || { echo "load calibration counters failed"; exit 1; }
timeout -k 10 60 $R/tests/_build/valu_calib m 20000 > $O/fma_mix_rate.txt 2>&1 || { echo "fma_mix rate failed"; exit 1; }
echo profiles done
