"""One denoising step of ONE chain as the GPU ran it: every kernel of the step in order with its duration and the gap in front of it (rocprofv3
kernel trace of `bench.py --streams 1`).  usage (GPU box): python scripts/step_timeline.py  (writes under /tmp, prints the table)"""
import csv, glob, os, subprocess, sys, collections
out = "/tmp/step_tl"
subprocess.run(["rm", "-rf", out])
env = dict(os.environ, TMPDIR="/tmp")
subprocess.run(["rocprofv3", "--kernel-trace", "-d", out, "-o", "t", "--output-format", "csv", "--", sys.executable, "bench.py"] + (sys.argv[1:] if len(sys.argv) > 1 else ["--steps", "6", "--warmup", "2", "--streams", "1"]) + [
                "--no-cpu-baseline", "--no-counters"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# the last full step: between the last two predictor kernels
idx = [i for i, k in enumerate(ks) if "predictor_kernel" in k[0]]
a, b = idx[-2] + 1, idx[-1] + 1
step = ks[a:b]
t0 = ks[a - 1][2]
print(f"one step = {len(step)} kernels, {(step[-1][2] - t0) / 1e3:.1f} us from the previous step's last kernel end to this one's")
prev = t0
agg = collections.OrderedDict()
for name, s, e in step:
    short = name.split("(")[0].replace("void ", "").replace("mi::", "")[:60]
    d = agg.setdefault(short, [0, 0.0, 0.0])
    d[0] += 1; d[1] += (e - s) / 1e3; d[2] += max(0, s - prev) / 1e3
    prev = max(prev, e)
print(f"{'kernel':62s} {'calls':>5s} {'busy us':>9s} {'gaps in front us':>17s}")
for k, (n, busy, gap) in agg.items():
    print(f"{k:62s} {n:5d} {busy:9.1f} {gap:17.1f}")
print(f"{'total':62s} {sum(v[0] for v in agg.values()):5d} {sum(v[1] for v in agg.values()):9.1f} {sum(v[2] for v in agg.values()):17.1f}")
