"""The headline's concurrent chains as the GPU ran them: per kernel name, calls, mean / p90 duration and the mean gap between the previous
kernel's end ON THE SAME QUEUE and this kernel's start (rocprofv3 kernel trace of the default `bench.py` run, the last full steps only).
A kernel that waits for CU resources behind other chains' workgroups shows up as a long duration (first wave placed, the rest waiting) or a
long gap (nothing placed).  usage (GPU box): python scripts/chains_timeline.py [bench args...]"""
import collections
import csv
import glob
import os
import subprocess
import sys

out = "/tmp/chains_tl"
subprocess.run(["rm", "-rf", out])
env = dict(os.environ, TMPDIR="/tmp")
args = sys.argv[1:] if len(sys.argv) > 1 else ["--steps", "8", "--warmup", "3"]
subprocess.run(["rocprofv3", "--kernel-trace", "-d", out, "-o", "t", "--output-format", "csv", "--", sys.executable, "bench.py"] + args +
               ["--no-cpu-baseline", "--no-counters"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
print("columns:", ",".join(rows[0].keys()))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
# keep the last 40 % of the trace (steady state of the timed window)
t_lo = int(rows[0]["Start_Timestamp"]) + int(0.6 * (t_end - int(rows[0]["Start_Timestamp"])))
last_end = {}
agg = collections.OrderedDict()
for r in rows:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get(qkey, "0")
    gap = s - last_end[q] if q in last_end else 0
    last_end[q] = e
    if s < t_lo:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mi::", "")[:70]
    agg.setdefault(name, []).append(((e - s) / 1e3, gap / 1e3))
print(f"{'kernel':72s} {'calls':>6s} {'mean us':>8s} {'p90 us':>8s} {'gap us':>8s} {'sum ms':>8s}")
tot = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
    d = sorted(x for x, _ in v)
    g = sum(x for _, x in v) / len(v)
    tot += sum(d)
    print(f"{k:72s} {len(v):6d} {sum(d) / len(d):8.1f} {d[int(0.9 * (len(d) - 1))]:8.1f} {g:8.1f} {sum(d) / 1e3:8.2f}")
for k, v in agg.items():   # (MI_SKIP=8 launches every node chain twice: the second launch of each pair finds its weights in L2)
    if "node_c" in k and os.environ.get("MI_SKIP") == "8":
        d = [x for x, _ in v]
        print(f"  {k[:40]}: first-of-pair mean {sum(d[0::2]) / max(1, len(d[0::2])):.1f} us, second-of-pair mean {sum(d[1::2]) / max(1, len(d[1::2])):.1f} us")
print(f"queues: {len(last_end)}; window {(t_end - t_lo) / 1e6:.2f} ms; kernel time {tot / 1e3:.2f} ms")
