set -e
b() { python bench.py --workload $1 --dim $2 --rule $3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 D=$2 rule $3', round(d['ms_per_step'],4), 'dense', d['config']['dense_windows'], 'tasks', d['config']['sparse_tasks'])"; }
for w in yh_like rd_like tt_like cora reddit dense; do b $w 32 3; done
b dense 32 0; b dense 32 2
