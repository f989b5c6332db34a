#!/bin/bash
# Development tool: per-kernel average durations (rocprofv3 --kernel-trace --stats) of one kbench command.
# usage (on the GPU box): bash tools/ktrace.sh <tag> --config C4 --set --rounds 3 [--variants ...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/ktrace_$tag
rm -rf $out
rocprofv3 --kernel-trace --stats -d $out -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/kbench.py "$@" > $out.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-90s calls %5s  avg %9.1f us  total %6.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
