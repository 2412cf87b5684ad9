#!/bin/bash
# scripts/mfma_power.hip in its three modes with socket power / sclk polled through rocm-smi beside each run.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_mfma_power.log
: > $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/mfma_power scripts/mfma_power.hip 2>>$O || { cat $O; exit 1; }
for mode in ${MODES:-0 1 2}; do
  /tmp/mfma_power $mode 4 > /tmp/mp.txt 2>&1 &
  BP=$!
  sleep 1.2
  : > /tmp/pw.txt
  for i in $(seq 1 10); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power \(W\)" >> /tmp/pw.txt; kill -0 $BP 2>/dev/null || break; sleep 0.15; done
  wait $BP
  cat /tmp/mp.txt >> $O
  python - >> $O <<'PY'
import re
t = open("/tmp/pw.txt").read()
clk = [int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", t)]
pw = [float(x) for x in re.findall(r"Power \(W\): ([\d.]+)", t)]
n = min(len(clk), len(pw))
busy = [(c, p) for c, p in zip(clk[:n], pw[:n]) if p > 500]
print(f"   power {min(p for _, p in busy):.0f}-{max(p for _, p in busy):.0f} W, sclk {min(c for c, _ in busy)}-{max(c for c, _ in busy)} MHz over {len(busy)} polls" if busy else f"   no poll above 500 W: {pw} {clk}")
PY
done
cat $O
