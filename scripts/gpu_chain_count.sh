cd $GRAFT_REPO_ROOT
for rep in 1 2; do for st in 4 3 5 6 8; do timeout 600 python bench.py --steps 40 --warmup 5 --streams $st --no-cpu-baseline --no-counters 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chains=$st', round(d['value'],3), 'structures/s', round(d['ms_per_step'],3), 'ms/step')"; done; done
