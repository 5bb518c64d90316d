set -o pipefail
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sr -- python3 $ROOT/profiles/seal_refmix.py > $OUT/seal_refmix_under_rocprof.log 2>&1
cp $OUT/prof_sr/*/*kernel_stats.csv $OUT/seal_refmix_kernel_stats.csv
cp $OUT/prof_sr/*/*kernel_trace.csv $OUT/seal_refmix_kernel_trace.csv
rm -rf $OUT/prof_sr
