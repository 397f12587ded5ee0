#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of one 16-frame launch on scene_suzanne for each EXTRA flag set
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for X in "$@"; do
  (cd $R/raytracer_amd/csrc && make -s clean >/dev/null && make -s EXTRA="$X" 2>/dev/null) || { echo "build failed: $X"; continue; }
  T=$(echo "$X" | tr -c 'A-Za-z0-9' '_'); O=$R/gpurun_out/ab_write/$T; mkdir -p $O
  timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 $R/tests/harness/prof_run.py 16 suzanne > $O/w.log 2>&1 || { echo "pass failed"; exit 1; }
  echo "== EXTRA='$X'"; grep "^ms " $O/w.log
  python3 - $O <<'PY'
import csv, glob, sys
t = 0.0
for f in glob.glob(sys.argv[1] + "/w/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_persistent_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE": t += float(r["Counter_Value"])
print("WRITE_SIZE x 1024 = %.1f MB per frame" % (t * 1024 / 16 / 1e6))
PY
done
(cd $R/raytracer_amd/csrc && make -s clean >/dev/null && make -s 2>/dev/null)
