#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh            -> gpurun_out/prof_round/{reddit,alldense,dense}/<pass>/
# Passes are separate runs (kernel trace + stats; then one PMC set per run), as MI355X_MICROARCH.md
# prescribes.  Summarise afterwards with profiles/summarize.py.
export TMPDIR=/tmp
OUT=gpurun_out/prof_round
mkdir -p $OUT
run() {  # name, bench args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/trace -- python3 bench.py "$@" --steps 50 --warmup 5 --no-cpu-baseline > $OUT/$name/trace_bench.json 2> $OUT/$name/trace.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$name/fetch -- python3 bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$name/write -- python3 bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/$name/l2 -- python3 bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/$name/mfma -- python3 bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  echo "$name done"
}
for n in reddit_d128 alldense_d128 dense_d128 reddit_d32 reddit_d256 rd_like_d32 yh_like_d32 reddit_d128_bf16; do mkdir -p $OUT/$n; done
run reddit_d128
run alldense_d128 --workload alldense
run dense_d128 --workload dense
run reddit_d32 --dim 32
run reddit_d256 --dim 256
run rd_like_d32 --workload rd_like
run yh_like_d32 --workload yh_like
run reddit_d128_bf16 --dtype bf16
