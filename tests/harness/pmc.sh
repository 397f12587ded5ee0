#!/bin/bash
# usage: pmc.sh <outdir-name> [frames] [scene] -- kernel-trace + PMC passes of prof_run.py, one rocprofv3 run per pass
# (at most 8 SQ, 4 TCC, 4 TCP, 2 TA, 2 TD, 2 GRBM counters per pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; F=${2:-4}; S=${3:-suzanne}; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tests/harness/prof_run.py $F $S > $O/kt.log 2>&1 || { echo "kernel trace failed"; exit 1; }
N=1
run() { timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc$N -- python3 $R/tests/harness/prof_run.py $F $S > $O/pmc$N.log 2>&1 || { echo "pass $N failed: $*"; exit 1; }; N=$((N+1)); }
run SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE
run SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR
run TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run FETCH_SIZE
run WRITE_SIZE
run SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_VMEM
run TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
python3 $R/tests/harness/pmc_summary.py $O > $O/summary.json
echo pmc done: $1
