set -o pipefail
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tw -- python3 $ROOT/profiles/twowl_scaled.py > $OUT/twowl_scaled_under_rocprof.log 2>&1
cp $OUT/prof_tw/*/*kernel_stats.csv $OUT/twowl_scaled_kernel_stats.csv
rm -rf $OUT/prof_tw
head -40 $OUT/twowl_scaled_kernel_stats.csv | cut -c1-200
