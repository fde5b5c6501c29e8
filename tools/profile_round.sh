#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh            -> gpurun_out/prof_round/{reddit,alldense,dense}/<pass>/
# Passes are separate runs (kernel trace + stats; then one PMC set per run), as MI355X_MICROARCH.md
# prescribes.  Summarise afterwards with profiles/summarize.py.
export TMPDIR=/tmp
OUT=${PROF_OUT:-gpurun_out/prof_round}   # PROF_OUT / SKIP_MFMA=1: A/B runs of variants selected through the environment
mkdir -p $OUT
run() {  # name, bench args...
  name=$1; shift
  set -- "$@" --graph-cache /tmp/hcspmm_graph_$name
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/trace -- python3 bench.py "$@" --steps 50 --warmup 5 --no-cpu-baseline --no-pmc --no-sweep > $OUT/$name/trace_bench.json 2> $OUT/$name/trace.err
  timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$name/fetch -- python3 bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-sweep > /dev/null 2>&1
  timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$name/write -- python3 bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-sweep > /dev/null 2>&1
  timeout -k 10 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/$name/l2 -- python3 bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-sweep > /dev/null 2>&1
  [ -n "$SKIP_MFMA" ] || timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/$name/mfma -- python3 bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-sweep > /dev/null 2>&1
  rm -rf /tmp/hcspmm_graph_$name
  echo "$name done"
}
# usage: profile_round.sh [key ...]   (default: every workload below)
declare -A W=( [reddit_d128]="" [reddit_d32]="--dim 32" [reddit_d256]="--dim 256" [cora_d32]="--workload cora"
               [products_share_d256]="--workload products_share" [c5_share_d128]="--workload c5_share"
               [alldense_d128]="--workload alldense" [dense_d128]="--workload dense" [rd_like_d32]="--workload rd_like"
               [yh_like_d32]="--workload yh_like" [reddit_d128_bf16]="--dtype bf16"
               [community_d128]="--workload community" [community_loi_d128]="--workload community_loi"
               [community_d32]="--workload community --dim 32" [community_loi_d32]="--workload community_loi --dim 32" [rd_like_d22]="--workload rd_like --dim 22" )
KEYS="$@"
[ -z "$KEYS" ] && KEYS="reddit_d128 reddit_d32 reddit_d256 cora_d32 products_share_d256 c5_share_d128 alldense_d128 rd_like_d32 yh_like_d32 reddit_d128_bf16"
# (gpurun merges new files into an existing gpurun_out/: clear this workload's directory first, or a later summary mixes rounds)
for n in $KEYS; do rm -rf $OUT/$n; mkdir -p $OUT/$n; run $n ${W[$n]}; done
