"""Per-kernel HBM-side bytes of the fine-tune micro-step (bench.py --mode ft), from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; corrected as
the MI355X guide prescribes): dispatches, fetch and write bytes per dispatch and per timestep, sorted by bytes per timestep.
usage (on the GPU box): python scripts/ft_traffic_table.py [steps] > gpurun_out/r6_ft_traffic.md"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
extra = sys.argv[2:]
got, why = bench._counter_passes(["--mode", "ft", "--steps", str(steps), "--warmup", "1"] + extra, timeout_s=400)
if got is None:
    sys.exit(why)
T = steps + 1
rows = []
for k in set(got["FETCH_SIZE"]) | set(got["WRITE_SIZE"]):
    nf, f = got["FETCH_SIZE"].get(k, (0, 0.0))
    nw, w = got["WRITE_SIZE"].get(k, (0, 0.0))
    n = max(nf, nw)
    fb, wb = 2.0 * f * 1024.0, w * 1024.0
    rows.append((n * (fb + wb) / T, k[:90], n, fb, wb))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"# fine-tune micro-step: HBM-side bytes per kernel ({T} timesteps per pass; {bench.COUNTER_CORRECTION})\n")
print(f"total per timestep: {tot / 1e9:.2f} GB\n")
print("| kernel | dispatches / timestep | fetch MB / dispatch | write MB / dispatch | GB / timestep | % |")
print("|---|---|---|---|---|---|")
for b, k, n, fb, wb in rows[:45]:
    print(f"| `{k}` | {n / T:.1f} | {fb / 1e6:.1f} | {wb / 1e6:.1f} | {b / 1e9:.3f} | {100 * b / tot:.1f} |")
