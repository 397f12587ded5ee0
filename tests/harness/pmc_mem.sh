#!/bin/bash
# vector-memory front-end counters for the persistent kernel (TA: 2 per pass, TCP: 4 per pass, TD: 2)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; F=${2:-8}; mkdir -p $O
run() { timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $O/m$N -- python3 $R/tests/harness/prof_run.py $F > $O/m$N.log 2>&1 || echo "pass $N failed"; N=$((N+1)); }
N=1
run TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE
run TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE
run TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum GRBM_GUI_ACTIVE
run TD_TD_BUSY_sum TD_TC_STALL_sum SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
python3 $R/tests/harness/pmc_summary.py $O | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d['counters'].items():
    if 'persistent' in k:
        for a,b in sorted(v.items()):
            if not a.startswith('dispatches') and not a.endswith('_Count') and not a.endswith('_Size'): print('%-42s %.4g' % (a, b))
"
