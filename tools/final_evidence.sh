#!/bin/bash
# The end-of-round evidence in one go (run through gpurun from the repo root): GPU suite, the same suite under the forced-slices
# stress environment, the default bench line, and the LOI logs.  Everything lands under gpurun_out/final/.
O=gpurun_out/final; mkdir -p $O   # (do not pipe this script into head: a closed pipe ends it early)
python -m pytest tests -q -m gpu > $O/gpu_tests_final.log 2>&1; tail -1 $O/gpu_tests_final.log
HCSPMM_SLICE_THRESHOLD=4 HCSPMM_TINY_KERNEL_MIN_TASKS=1 python -m pytest tests -q -m gpu > $O/gpu_tests_forced_slices.log 2>&1; tail -1 $O/gpu_tests_forced_slices.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/show_bench.py $O/bench_default.json > $O/bench_default.txt 2>&1; head -2 $O/bench_default.txt
HCSPMM_LOI_DEBUG=1 python tools/loi_fast_timing.py > $O/loi_fast_timing.log 2>&1; echo "loi timing done"
for m in off node l3; do echo "== HCSPMM_LOI_PIN=$m"; HCSPMM_LOI_PIN=$m LOI_EXACT=0 python tools/loi_fast_timing.py community 2>&1 | grep -v "^loi_fast\|amdgpu"; done > $O/loi_pin.log 2>&1; echo "pin done"
python tools/loi_pipeline_experiment.py > $O/loi_pipeline.log 2>&1; echo "pipeline done"
