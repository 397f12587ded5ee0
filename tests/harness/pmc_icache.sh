#!/bin/bash
# instruction-cache and instruction-fetch counters of one 16-frame launch (scene_suzanne 1080p x 16 spp)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3/icache; mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" $O/avail.txt | sort -u | tr '\n' ' '
echo
timeout -s KILL 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 $R/tests/harness/prof_run.py 16 suzanne > $O/p1.log 2>&1 || { echo "pass failed"; tail -5 $O/p1.log; exit 1; }
python3 - $O/p1 <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_persistent" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
print(dict(tot))
PY
