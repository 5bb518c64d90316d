#!/bin/bash
# SQ counters of one kernel: bash profiles/pmc_kernel.sh <out dir under gpurun_out> <kernel regex> <script> [args...]
# four counter-only passes (no traces; the last is GRBM_GUI_ACTIVE: / 8 / the kernel's wall time = the clock it ran at), then per-launch averages into <out>/summary.txt
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; RE=$2; shift 2
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-include-regex "$RE" --output-format csv -d $OUT/pass1 -- python3 "$@" > $OUT/pass1.log 2>&1 &&
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES --kernel-include-regex "$RE" --output-format csv -d $OUT/pass2 -- python3 "$@" > $OUT/pass2.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT --kernel-include-regex "$RE" --output-format csv -d $OUT/pass3 -- python3 "$@" > $OUT/pass3.log 2>&1 &&
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-include-regex "$RE" --output-format csv -d $OUT/pass4 -- python3 "$@" > $OUT/pass4.log 2>&1 &&
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "pass*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(sys.argv[1], "summary.txt"), "w") as out:
    for k, cs in acc.items():
        print(k, file=out)
        for c in sorted(cs):
            v = cs[c]
            print(f"  {c:28s} {sum(v) / len(v):14.4g}   ({len(v)} launches)", file=out)
PY
cat $OUT/summary.txt
