#!/bin/bash
cd $GRAFT_REPO_ROOT
for g in 4 6 8 4 6 8; do echo -n "groups=$g: "; python bench.py --mode ft --ft-groups $g --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*, "unit": "crystal-timesteps[^,]*'; done
