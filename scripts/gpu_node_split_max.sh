cd $GRAFT_REPO_ROOT
for rep in 1 2; do for f in 256 512 100000; do
for st in 1 2; do MI_NODE_SPLIT=$f timeout 600 python bench.py --steps 40 --warmup 5 --streams $st --no-cpu-baseline --no-counters 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('split_max=$f chains=$st', round(d['value'],3), 'structures/s', round(d['ms_per_step'],3), 'ms/step')"; done
done; done
