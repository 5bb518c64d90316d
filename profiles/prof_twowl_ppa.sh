set -o pipefail
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_twp -- python3 $ROOT/profiles/twowl_ppa.py > $OUT/twowl_ppa_under_rocprof.log 2>&1
cp $OUT/prof_twp/*/*kernel_stats.csv $OUT/twowl_ppa_kernel_stats.csv
rm -rf $OUT/prof_twp
