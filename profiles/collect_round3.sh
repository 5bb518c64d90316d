#!/bin/bash
# Round-3 measurement artefacts on the GPU box:  gpurun -- 'bash profiles/collect_round3.sh'
# everything lands under gpurun_out/r03/ (copied to profiles/r03/ afterwards); counter passes are counter-only runs.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
echo "[collect] bench done"
python profiles/twowl_scaled.py > $OUT/twowl_scaled.log 2>&1
python profiles/twowl_ppa.py > $OUT/twowl_ppa.log 2>&1
python profiles/seal_refmix.py > $OUT/seal_refmix.log 2>&1
python profiles/seal_stages.py > $OUT/seal_stages.log 2>&1
python profiles/seal_extract_probe.py > $OUT/seal_extract_by_size.log 2>&1
LPG_LIB_VARIANT=tuning python profiles/seal_phases.py > $OUT/seal_phases.log 2>&1
python profiles/spmm_mid_graph.py > $OUT/spmm_mid_graph.log 2>&1
python profiles/degree_ones_probe.py > $OUT/degree_ones_probe.log 2>&1
python profiles/kernel_zoo.py > $OUT/kernel_zoo.log 2>&1
python profiles/linear_x3_tiled.py > $OUT/linear_x3_tiled.log 2>&1
python profiles/linear_bf16_shapes.py > $OUT/linear_bf16_shapes.log 2>&1
LPG_LIB_VARIANT=tuning python profiles/seal_conv_phases.py > $OUT/seal_conv_phases.log 2>&1
echo "[collect] scripts done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $ROOT/bench.py > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tw -- python3 $ROOT/profiles/twowl_scaled.py > $OUT/twowl_scaled_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sr -- python3 $ROOT/profiles/seal_refmix.py > $OUT/seal_refmix_under_rocprof.log 2>&1
echo "[collect] traces done"
cd $ROOT
cp $OUT/prof_bench/*/*kernel_stats.csv $OUT/bench_kernel_stats.csv
cp $OUT/prof_tw/*/*kernel_stats.csv $OUT/twowl_scaled_kernel_stats.csv
cp $OUT/prof_sr/*/*kernel_stats.csv $OUT/seal_refmix_kernel_stats.csv
rm -rf $OUT/prof_bench $OUT/prof_tw $OUT/prof_sr $OUT/seal_refmix_kernel_trace.csv
bash profiles/collect_traffic.sh > $OUT/collect_traffic.log 2>&1
cp gpurun_out/traffic/traffic_latest.json $OUT/ 2>/dev/null
cp gpurun_out/traffic/csv/*.csv $OUT/ 2>/dev/null
ls -la $OUT
