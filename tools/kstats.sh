#!/bin/bash
# Runs on the GPU box: per-kernel times of one tool run (rocprofv3 --kernel-trace --stats), top kernels printed.  usage: tools/kstats.sh <name> <script> [args]
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/kstats_$NAME" -- python3 "$GRAFT_REPO_ROOT/$@" > "$GRAFT_REPO_ROOT/gpurun_out/kstats_$NAME.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python3 - "$NAME" <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/kstats_%s/**/*kernel_stats.csv' % sys.argv[1], recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print("%-90s calls %5s avg %10.1f us" % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3))
PY
