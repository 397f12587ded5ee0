#!/bin/bash
# usage: pmc.sh <outdir-name> [frames] -- runs kernel-trace + PMC passes of prof_run.py
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; F=${2:-4}; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tests/harness/prof_run.py $F > $O/kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $O/pmc1 -- python3 $R/tests/harness/prof_run.py $F > $O/pmc1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc2 -- python3 $R/tests/harness/prof_run.py $F > $O/pmc2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc3 -- python3 $R/tests/harness/prof_run.py $F > $O/pmc3.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc4 -- python3 $R/tests/harness/prof_run.py $F > $O/pmc4.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc5 -- python3 $R/tests/harness/prof_run.py $F > $O/pmc5.log 2>&1
echo pmc done
