#!/bin/bash
# Re-collect profiles/traffic_latest.json after a change to csrc/spmm.hip (bench.py reports roofline.traffic only while
# the file's sha256 matches):  gpurun -- 'bash profiles/collect_traffic.sh', then copy gpurun_out/traffic/traffic_latest.json
# to profiles/ and gpurun_out/traffic/csv/*.csv to profiles/<round>/.  Two counter-only passes (no traces).
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/profiles/pmc_workload.py > $OUT/fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/profiles/pmc_workload.py > $OUT/write.log 2>&1 &&
cd $GRAFT_REPO_ROOT && python profiles/make_traffic.py $OUT/pmc_fetch $OUT/pmc_write gpurun_out/traffic/csv && cp profiles/traffic_latest.json gpurun_out/traffic/ && cat profiles/traffic_latest.json | head -30
