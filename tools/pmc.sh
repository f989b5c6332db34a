#!/bin/bash
# Runs on the GPU box: one rocprofv3 --pmc pass per counter group over a tool run; per-kernel sums printed.  usage: tools/pmc.sh <name> "<counters>" <script> [args]
NAME=$1; CTR=$2; shift 2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$NAME" -- python3 "$GRAFT_REPO_ROOT/$@" > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$NAME.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python3 - "$NAME" <<'PY'
import csv, glob, sys, collections
f = glob.glob('gpurun_out/pmc_%s/**/*counter_collection.csv' % sys.argv[1], recursive=True)
if not f: sys.exit('no counters collected: see gpurun_out/pmc_%s.log' % sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'][:50]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    calls[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    if 'acgpu' not in k: continue
    print(k, {c: "%.4g per call" % (v / calls[(k, c)]) for c, v in d.items()})
PY
