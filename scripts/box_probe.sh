#!/bin/bash
# What the box's GPU runs at WHILE the headline loop runs: the amdgpu hwmon / sysfs files sampled every 20 ms beside a timed bench.
# usage (through gpurun): bash scripts/box_probe.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-probe}; shift
O=gpurun_out/r5_box_$TAG.log
D=$(ls -d /sys/class/drm/card*/device | head -1)
H=$(ls -d $D/hwmon/hwmon* 2>/dev/null | head -1)
{ echo "device $D hwmon $H"; ls $D | tr '\n' ' '; echo; ls $H 2>/dev/null | tr '\n' ' '; echo;
  for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk power_dpm_force_performance_level; do echo "== $f"; cat $D/$f 2>/dev/null; done;
  for f in $H/*_label $H/power1_cap $H/power1_cap_max; do echo "$f: $(cat $f 2>/dev/null)"; done; } > $O 2>&1
python - "$H" "$D" $O.samples "$@" <<'PY' >> $O 2>&1
import sys, time, threading, subprocess, glob, os
H, D, out = sys.argv[1], sys.argv[2], sys.argv[3]
args = sys.argv[4:]
files = {}
for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input", "temp2_input", "temp3_input"):
    p = os.path.join(H, name)
    if os.path.exists(p): files[name] = p
stop = False
rows = []
def sample():
    while not stop:
        r = [time.time()]
        for n, p in files.items():
            try: r.append(int(open(p).read().strip()))
            except Exception: r.append(-1)
        rows.append(r)
        time.sleep(0.02)
t = threading.Thread(target=sample); t.start()
t0 = time.time()
p = subprocess.run([sys.executable, "bench.py"] + args, capture_output=True, text=True)
stop = True; t.join()
print("bench rc", p.returncode, "wall", round(time.time() - t0, 1))
print(p.stdout[-3000:]); print(p.stderr[-1500:])
names = list(files)
with open(out, "w") as f:
    f.write("t " + " ".join(names) + "\n")
    for r in rows: f.write(" ".join(str(x) for x in r) + "\n")
import statistics
for i, n in enumerate(names):
    v = [r[i + 1] for r in rows if r[i + 1] >= 0]
    if v:
        vs = sorted(v)
        print(n, "min", vs[0], "p10", vs[len(vs) // 10], "median", vs[len(vs) // 2], "p90", vs[9 * len(vs) // 10], "max", vs[-1], "n", len(v))
PY
tail -40 $O
