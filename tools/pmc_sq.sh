#!/bin/bash
# Runs on the GPU box: SQ instruction / busy counters of the tile kernel variants (counters only, no other trace domain).
# usage: bash tools/pmc_sq.sh <out-dir under gpurun_out/> '<kbench variants json>' ['<more kbench args>' [kernel-name substring]]
set -u
OUT=gpurun_out/${1:-pmc_sq}
V=${2:-'{"tile":{}}'}
KARGS=${3:-}
export KPAT=${4:-k_ac_tile}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d "$OUT/a" -- python3 tools/kbench.py --rounds 1 $KARGS --variants "$V" > "$OUT/a.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/b" -- python3 tools/kbench.py --rounds 1 $KARGS --variants "$V" > "$OUT/b.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for sub in ("a", "b"):
    files = glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if os.environ.get("KPAT", "k_ac_tile") not in k: continue
            agg[(k, r["Dispatch_Id"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, d), cs in sorted(agg.items(), key=lambda kv: int(kv[0][1])):
        print(sub, d, k[:70], {c: round(sum(v) / 1e6, 2) for c, v in cs.items()})
PY
