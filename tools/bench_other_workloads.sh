#!/bin/bash
# Plain bench lines for the workloads that are not in the default sweep: the TT / YeastH / DP-sized graphs of the paper's Table II and BASELINE configs 4 and 5 at
# FULL size on one GPU (run through gpurun from the repo root; the digest goes to profiles/rNN/bench_other_workloads.log).
for w in tt_like yh_like dp_like; do echo "== $w"; python bench.py --workload $w --no-sweep --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps({k:d[k] for k in ('value','ms_per_step')}), 'kernel_ms', round(r['kernel_ms'],4), 'traffic GB', round((r.get('traffic') or 0)/1e9,3), 'frac', round(r.get('frac',0),3), 'dense_windows', d['config']['dense_windows'], 'prep ms', round(d['config']['preprocess_ms'],1))"; done
for w in products c5; do echo "== $w (full size, one GPU)"; python bench.py --workload $w --no-cpu-baseline --steps 30 --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps({k:d[k] for k in ('value','ms_per_step','scaling')}), 'kernel_ms', round(r['kernel_ms'],4), 'dense_windows', d['config']['dense_windows'], 'prep ms', round(d['config']['preprocess_ms'],1))"; done
