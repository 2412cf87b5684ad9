#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for st in 2 3 4; do echo -n "streams=$st: "; python bench.py --steps 100 --warmup 3 --streams $st --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*, "unit": "struct[^,]*'; done; done
