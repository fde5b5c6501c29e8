#!/bin/bash
# A/B of library variants on one box: tools/ab_libs.sh "name=ENV..:LIB ..." "workload[:dim] ..."   (LIB empty = the in-tree build)
#   e.g. tools/ab_libs.sh "off=HCSPMM_TINY_KERNEL_MIN_TASKS=-1 t4w8= t4w6=:_ab_libs/t4w6.so" "rd_like yh_like tt_like:32"
# EXTRA="--dtype bf16" etc. is appended to every bench.py command.
# One bench.py run per (variant, workload): headline only, no counter passes; prints kernel time (HIP events) per step.
VARIANTS=$1; WORKLOADS=$2; STEPS=${STEPS:-100}
for w in $WORKLOADS; do
  wl=${w%%:*}; dim=""; [ "$w" != "$wl" ] && dim="--dim ${w##*:}"
  for v in $VARIANTS; do
    name=${v%%=*}; rest=${v#*=}; envs=${rest%%:*}; lib=""; [ "$rest" != "$envs" ] && lib=${rest##*:}
    out=$(env ${envs:+$envs} ${lib:+HCSPMM_LIB=$PWD/$lib} python3 bench.py --workload $wl $dim --steps $STEPS --warmup 10 --no-sweep --no-pmc --no-cpu-baseline --frontend ctypes $EXTRA 2>/dev/null)
    python3 - "$w" "$name" <<PY
import json,sys
try:
    d=json.loads('''$out'''.strip().splitlines()[-1])
    print("%-22s %-8s kernel %8.4f ms  step %8.4f ms  %6.3fe12 edge*dim/s" % (sys.argv[1], sys.argv[2], d["roofline"]["kernel_ms"], d["ms_per_step"], d["value"]/1e12), flush=True)
except Exception as e:
    print("%-22s %-8s FAILED %s" % (sys.argv[1], sys.argv[2], e), flush=True)
PY
  done
done
