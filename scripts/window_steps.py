"""Per-chain, per-step durations INSIDE the driver's 20-step window (rocprofv3 kernel trace of the driver's command): are the first steps of a sample() call
slower than the steady state, and how ragged are the ends?  usage (GPU box): python scripts/window_steps.py [steps] [warmup]"""
import csv, glob, os, subprocess, sys, collections
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
out = "/tmp/window_steps"
subprocess.run(["rm", "-rf", out])
env = dict(os.environ, TMPDIR="/tmp")
cmd = [sys.executable, "bench.py", "--steps", str(K), "--warmup", str(W), "--no-cpu-baseline", "--no-counters"] + sys.argv[3:]
if os.environ.get("MI_WINDOW_CMD"):   # (another workload whose last K predictor kernels per queue are to be split into steps, e.g. scripts/two_calls.py)
    cmd = [sys.executable] + os.environ["MI_WINDOW_CMD"].split()
subprocess.run(["rocprofv3", "--kernel-trace", "-d", out, "-o", "t", "--output-format", "csv", "--"] + cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
byq = collections.defaultdict(list)
for r in rows:
    byq[r[qkey]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
chains = []
for q, ks in byq.items():
    ks.sort()
    pred = [(s, e) for s, e, n in ks if "predictor_kernel" in n]
    if len(pred) >= K:
        chains.append((q, ks, pred))
print(f"{len(chains)} queues carry predictor kernels; the last {K} of each are the timed window")
t_lo = min(p[-K - 1][1] if len(p) > K else p[-K][0] for _, _, p in chains)   # end of the step in front of the window (warm-up's last predictor)
firsts = []
for q, ks, pred in chains:
    win = pred[-K:]
    prev_end = pred[-K - 1][1] if len(pred) > K else None
    # the window's first kernel on this queue: the first kernel after the warm-up call's last predictor (+ the zero-step call's kernels, if any)
    start = min(s for s, e, n in ks if prev_end is None or s > prev_end)
    ends = [e for s, e in win]
    durs = [(ends[0] - start) / 1e3] + [(ends[i] - ends[i - 1]) / 1e3 for i in range(1, K)]
    firsts.append(start)
    # busy time (sum of this queue's kernel durations) and the three big kernels' mean durations, step by step
    bounds = [start] + ends
    busy, big = [], []
    for i in range(K):
        kk = [(s, e, n) for s, e, n in ks if bounds[i] <= s and e <= bounds[i + 1] + 1]
        busy.append(sum(e - s for s, e, n in kk) / 1e3)
        g1 = [e - s for s, e, n in kk if "edge_gemm1b" in n]
        g2 = [e - s for s, e, n in kk if "edge_gemm2b" in n]
        nc = [e - s for s, e, n in kk if "node_chain" in n]
        big.append((sum(g1) / max(1, len(g1)) / 1e3, sum(g2) / max(1, len(g2)) / 1e3, sum(nc) / max(1, len(nc)) / 1e3, len(kk)))
    if os.environ.get("MI_DUMP_STEP1") and q == os.environ["MI_DUMP_STEP1"]:
        prev = None
        for s_, e_, n_ in [(s, e, n) for s, e, n in ks if bounds[0] <= s and e <= bounds[1] + 1]:
            print(f"            +{(s_ - start) / 1e3:8.1f} us  dur {(e_ - s_) / 1e3:7.1f}  gap {0.0 if prev is None else (s_ - prev) / 1e3:7.1f}  {n_.split('(')[0].replace('void ', '').replace('mi::', '')[:50]}")
            prev = e_
    print("          busy us per step: " + " ".join(f"{b:.0f}" for b in busy))
    print("          mean gemm1b / gemm2b / node_chain us, kernels: " + " | ".join(f"{a:.0f} {b:.0f} {c:.0f} ({n})" for a, b, c, n in big[:6]) + " ... " + " | ".join(f"{a:.0f} {b:.0f} {c:.0f} ({n})" for a, b, c, n in big[10:12]))
    print(f"queue {q}: first kernel at +{(start - min(firsts + [start])) / 1e3:.0f} us; step durations (us): " + " ".join(f"{d:.0f}" for d in durs))
    print(f"          mean of steps 1-3 {sum(durs[:3]) / 3:.0f}, steps 4-{K - 3} {sum(durs[3:K - 3]) / max(1, K - 6):.0f}, last 3 {sum(durs[-3:]) / 3:.0f}; chain ends at +{(ends[-1] - min(firsts)) / 1e3:.0f} us")
t0 = min(firsts)
t1 = max(p[-1][1] for _, _, p in chains)
print(f"window on the GPU: {(t1 - t0) / 1e3:.0f} us = {(t1 - t0) / 1e3 / K:.1f} us per step")
