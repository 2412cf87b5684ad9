#!/bin/bash
# Which part of the step draws the power: socket power and average sclk (rocm-smi, 150 ms polls) while the headline runs with parts of the forward skipped
# (mi_debug_set_skip: timing / power ablation only, results garbage).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_box_power_skip.log
: > $O
for sk in 0 1 2 4 6 3 5; do
  MI_SKIP=$sk python bench.py --steps 1000 --warmup 5 --no-cpu-baseline --no-counters $EXTRA > /tmp/pw.json 2>/dev/null &
  BP=$!
  sleep 3.2
  : > /tmp/pw.txt
  for i in $(seq 1 8); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power \(W\)" >> /tmp/pw.txt; kill -0 $BP 2>/dev/null || break; sleep 0.15; done
  wait $BP
  python - "$sk" >> $O <<'PY'
import re, sys, json
t = open("/tmp/pw.txt").read()
clk = [int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", t)]
pw = [float(x) for x in re.findall(r"Power \(W\): ([\d.]+)", t)]
n = min(len(clk), len(pw))
busy = [(c, p) for c, p in zip(clk[:n], pw[:n]) if p > 600]
d = json.loads(open("/tmp/pw.json").read().strip().splitlines()[-1])
names = {0: "full step", 1: "no node chain", 2: "no first edge GEMM", 4: "no second edge GEMM", 6: "neither edge GEMM", 3: "no node chain, no first edge GEMM", 5: "no node chain, no second edge GEMM"}
if busy:
    print(f"MI_SKIP={sys.argv[1]} ({names[int(sys.argv[1])]}): {d['ms_per_step']:.3f} ms/step; power {min(p for _, p in busy):.0f}-{max(p for _, p in busy):.0f} W, sclk {min(c for c, _ in busy)}-{max(c for c, _ in busy)} MHz over {len(busy)} polls")
else:
    print(f"MI_SKIP={sys.argv[1]} ({names[int(sys.argv[1])]}): {d['ms_per_step']:.3f} ms/step; no poll above 600 W: power {pw}, sclk {clk}")
PY
done
cat $O
