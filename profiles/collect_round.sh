#!/bin/bash
# Collect a round's measurement artefacts on the GPU box:  bash profiles/collect_round.sh r02
# (run through gpurun; everything lands under gpurun_out/<round>/ and is copied to profiles/<round>/ afterwards)
set -o pipefail
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
python profiles/seal_stages.py > $OUT/seal_stages.log 2>&1
python profiles/seal_stages.py --variant 1 2>&1 | grep "fp32:" | sed 's/^/option seal_conv_variant = 1 (the round-1 conv kernel): /' >> $OUT/seal_stages.log
python profiles/seal_extract_probe.py > $OUT/seal_extract_by_size.log 2>&1
python profiles/seal_fb_structure.py > $OUT/seal_fb_structure.log 2>&1
python profiles/degree_ab.py > $OUT/degree_ab.log 2>&1
python profiles/twowl_forward.py > $OUT/twowl_forward.log 2>&1
python profiles/kernel_zoo.py > $OUT/kernel_zoo_after.log 2>&1
python profiles/scatter_crossover.py > $OUT/scatter_crossover.log 2>&1
python profiles/linear_x3.py > $OUT/linear_x3.log 2>&1
python profiles/seal_pipeline.py > $OUT/seal_pipeline.log 2>&1
python profiles/config_sweep.py > $OUT/config4_config5_shapes.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $ROOT/bench.py > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_seal -- python3 $ROOT/profiles/seal_stages.py > $OUT/seal_stages_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_twowl -- python3 $ROOT/profiles/twowl_forward.py > $OUT/twowl_under_rocprof.log 2>&1
cd $ROOT
cp $OUT/prof_bench/*/*kernel_stats.csv $OUT/bench_kernel_stats.csv
cp $OUT/prof_seal/*/*kernel_stats.csv $OUT/seal_kernel_stats.csv
cp $OUT/prof_twowl/*/*kernel_stats.csv $OUT/twowl_kernel_stats.csv
rm -rf $OUT/prof_bench $OUT/prof_seal $OUT/prof_twowl
ls -la $OUT
