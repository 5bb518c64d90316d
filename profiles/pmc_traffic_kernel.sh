#!/bin/bash
# HBM-side traffic of one kernel: bash profiles/pmc_traffic_kernel.sh <out dir under gpurun_out> <kernel regex> <script> [args...]
# two counter-only passes (FETCH_SIZE, WRITE_SIZE: KiB per dispatch; FETCH_SIZE x 2 on gfx950, see profiles/make_traffic.py)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; RE=$2; shift 2
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$RE" --output-format csv -d $OUT/fetch -- python3 "$@" > $OUT/fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$RE" --output-format csv -d $OUT/write -- python3 "$@" > $OUT/write.log 2>&1 &&
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(sys.argv[1], "traffic.txt"), "w") as out:
    for k, cs in acc.items():
        f = sum(cs["FETCH_SIZE"]) / max(1, len(cs["FETCH_SIZE"])) * 1024 * 2
        w = sum(cs["WRITE_SIZE"]) / max(1, len(cs["WRITE_SIZE"])) * 1024
        print(f"{k}: fetch {f / 1e9:.3f} GB (FETCH_SIZE x 2), write {w / 1e9:.3f} GB per launch ({len(cs['FETCH_SIZE'])} launches)", file=out)
PY
cat $OUT/traffic.txt
