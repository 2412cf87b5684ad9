#!/bin/bash
# shader clock and package power while (a) a 4-crystal chain, (b) the headline batch is running
cd $GRAFT_REPO_ROOT
(for i in 1 2 3 4 5 6 7 8; do python scripts/small_batch_step.py 10,7,4,10 990; done > /tmp/sb.log 2>&1) &
PID=$!
sleep 14
for i in 1 2 3; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2; rocm-smi --showpower 2>/dev/null | grep -i "power (W)" | head -1; sleep 1; done
kill $PID 2>/dev/null; wait $PID 2>/dev/null; tail -2 /tmp/sb.log
(python bench.py --steps 400 --warmup 3 --no-cpu-baseline > /tmp/b.log 2>&1) &
PID=$!
sleep 14
for i in 1 2 3; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2; rocm-smi --showpower 2>/dev/null | grep -i "power (W)" | head -1; sleep 0.5; done
wait $PID; cut -c1-160 /tmp/b.log
