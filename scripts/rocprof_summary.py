#!/usr/bin/env python
"""Summarise rocprofv3 rocpd databases (kernel trace + PMC passes) into a small markdown file.
usage: rocprof_summary.py OUT.md TRACE.db [PMC.db ...]
If a pass holds FETCH_SIZE / WRITE_SIZE, OUT.md's sibling OUT_traffic.json receives the per-dispatch HBM-side bytes of the
dominant kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: the counters are in KiB, and on
gfx950 FETCH_SIZE reports half the bytes of wide coalesced streaming reads (doubled here); WRITE_SIZE is taken as reported."""
import json
import sqlite3
import sys


def main():
    out, trace, pmcs = sys.argv[1], sys.argv[2], sys.argv[3:]
    lines = ["# rocprofv3 summary", "", f"source: `{trace}` (+ {len(pmcs)} PMC passes)", "",
             "## kernel trace (`--kernel-trace --stats`)", "", "| kernel | calls | total us | avg us | % |", "|---|---|---|---|---|"]
    cur = sqlite3.connect(trace).cursor()
    rows = [r for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels")
            if "spin_kernel" not in r[0]]   # (the stream-concurrency probe runs once at start-up, outside the timed region)
    tot_all = sum(r[2] for r in rows) or 1.0
    for name, calls, tot, avg, pct in rows[:int(__import__("os").environ.get("MI_SUMMARY_ROWS", "16"))]:
        lines.append(f"| `{name[:110]}` | {calls} | {tot:.1f} | {avg:.2f} | {100.0 * tot / tot_all:.2f} |")
    for p in pmcs:
        cur = sqlite3.connect(p).cursor()
        lines += ["", f"## PMC pass `{p}` (per-dispatch averages)", "", "| kernel | counter | dispatches | avg value |", "|---|---|---|---|"]
        q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
             "group by kernel_name, counter_name order by sum(value) desc")
        rows = [r for r in cur.execute(q) if any(t in r[0] for t in ("edge_mlp", "gemm_nt", "gemm_planes", "edge_gemm", "node_chain"))]
        for k, c, n, v in rows:
            lines.append(f"| `{k[:60]}` | {c} | {n} | {v:.6g} |")
    # HBM-side traffic and achieved bandwidth of every kernel (north_star: "achieved HBM GB/s against the chip's peak")
    dur = {}
    cur = sqlite3.connect(trace).cursor()
    for name, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        dur[name.split("(")[0]] = (avg, calls)   # avg in us
    per = {}
    for p in pmcs:
        cur = sqlite3.connect(p).cursor()
        for k, c, v in cur.execute("select kernel_name, counter_name, avg(value) from counters_collection where counter_name in "
                                   "('FETCH_SIZE', 'WRITE_SIZE') group by kernel_name, counter_name"):
            per.setdefault(k.split("(")[0], {})[c] = v
    rows = []
    for k, d in per.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d and k in dur:
            fetch, write = 2.0 * d["FETCH_SIZE"] * 1024.0, d["WRITE_SIZE"] * 1024.0
            avg_us = dur[k][0]
            rows.append((dur[k][0] * dur[k][1], k, avg_us, fetch, write, (fetch + write) / (avg_us * 1e-6) / 1e9))
    if rows:
        lines += ["", "## HBM-side traffic per dispatch (FETCH_SIZE x2 correction, WRITE_SIZE as reported) and achieved bandwidth", "",
                  "Durations from the kernel trace, bytes from the PMC passes of the same command; peak 8000 GB/s.", "",
                  "| kernel | avg us | fetch MB | write MB | GB/s | % of peak |", "|---|---|---|---|---|---|"]
        for _, k, us, f, w, gbs in sorted(rows, reverse=True)[:14]:
            lines.append(f"| `{k[:70]}` | {us:.1f} | {f / 1e6:.1f} | {w / 1e6:.1f} | {gbs:.0f} | {gbs / 80:.1f} |")
        # the whole trace: every kernel that appears in all three passes, weighted by its calls (north_star: the sampler's
        # achieved HBM GB/s against the chip's peak)
        tot_b = sum((f + w) * dur[k][1] for _, k, us, f, w, gbs in rows)
        tot_us = sum(us * dur[k][1] for _, k, us, f, w, gbs in rows)
        all_us = sum(a * c for a, c in dur.values())
        whole = {"bytes": tot_b, "kernel_time_us": tot_us, "GBps": tot_b / (tot_us * 1e-6) / 1e9, "frac_of_8TBps": tot_b / (tot_us * 1e-6) / 8e12,
                 "share_of_trace_time_covered": tot_us / all_us}
        lines += ["", f"Whole trace (kernels present in all passes, {100 * whole['share_of_trace_time_covered']:.1f} % of the kernel time): "
                      f"{tot_b / 1e9:.2f} GB in {tot_us / 1e3:.1f} ms of kernel time = {whole['GBps']:.0f} GB/s = {100 * whole['frac_of_8TBps']:.1f} % of the 8 TB/s peak"]
    whole = locals().get("whole")
    traffic = {}
    for p in pmcs:
        cur = sqlite3.connect(p).cursor()
        # the two kernels of one bench 'launch' (pair-mode Fourier GEMM + second-linear GEMM): equally many dispatches each,
        # so the average over both x 2 = bytes per launch
        # round 3: the edge stage of an inference forward = gemm_planes_kernel<1, ...> (pair mode) + edge_gemm2b_kernel (edge_stage.hip), one
        # dispatch of each per layer; the node-level products run inside node_chain_kernel.  Older traces: both on gemm_planes kernels.
        has_e2 = cur.execute("select count(*) from counters_collection where kernel_name like '%edge_gemm2%'").fetchone()[0] > 0
        where = ("(kernel_name like '%gemm_planes_kernel<1%' or kernel_name like '%edge_gemm2%' or kernel_name like '%edge_gemm1%')" if has_e2 else
                 "(kernel_name like '%gemm_planes_db%' or kernel_name like '%gemm_planes_kernel%')")
        q = f"select counter_name, count(*), avg(value) from counters_collection where {where} and counter_name in ('FETCH_SIZE', 'WRITE_SIZE') group by counter_name"
        for c, n, v in cur.execute(q):
            traffic[c] = {"dispatches": n, "avg_reported_KiB": v}
        kernel_desc = ("gemm_planes_kernel<1> (pair mode) + edge_gemm2b_kernel (128-row register tiles, segmented sum on the matrix pipe): the edge stage of one layer"
                       if has_e2 else None)
    if "FETCH_SIZE" in traffic and "WRITE_SIZE" in traffic:
        fetch = 2.0 * traffic["FETCH_SIZE"]["avg_reported_KiB"] * 1024.0
        write = traffic["WRITE_SIZE"]["avg_reported_KiB"] * 1024.0
        rec = {"kernel": locals().get("kernel_desc") or "gemm_planes_kernel<1> (pair mode) + gemm_planes_kernel<0> (edge stage; the FETCH/WRITE passes run with the node-level products on the fp32-operand kernel so that every plane-GEMM dispatch is an edge-stage one)", "bytes_per_dispatch": fetch + write, "fetch_bytes_per_dispatch": fetch,
               "write_bytes_per_dispatch": write, "dispatches_per_bench_launch": 2, "counters": traffic,
               "correction": "KiB -> bytes; FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE as reported"}
        if whole:
            rec["whole_trace"] = whole
        json.dump(rec, open(out.rsplit(".", 1)[0] + "_traffic.json", "w"), indent=1)
        lines += ["", f"HBM-side traffic of the edge-stage `gemm_planes_kernel` dispatches, per dispatch (corrected): fetch {fetch / 1e6:.1f} MB + write {write / 1e6:.1f} MB"]
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
