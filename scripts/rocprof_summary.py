#!/usr/bin/env python
"""Summarise rocprofv3 rocpd databases (kernel trace + PMC passes) into a small markdown file.
usage: rocprof_summary.py OUT.md TRACE.db [PMC.db ...]"""
import sqlite3
import sys


def main():
    out, trace, pmcs = sys.argv[1], sys.argv[2], sys.argv[3:]
    lines = ["# rocprofv3 summary", "", f"source: `{trace}` (+ {len(pmcs)} PMC passes)", "",
             "## kernel trace (`--kernel-trace --stats`)", "", "| kernel | calls | total us | avg us | % |", "|---|---|---|---|---|"]
    cur = sqlite3.connect(trace).cursor()
    for name, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 16"):
        lines.append(f"| `{name[:110]}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} |")
    for p in pmcs:
        cur = sqlite3.connect(p).cursor()
        lines += ["", f"## PMC pass `{p}` (per-dispatch averages)", "", "| kernel | counter | dispatches | avg value |", "|---|---|---|---|"]
        q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
             "group by kernel_name, counter_name order by sum(value) desc")
        rows = [r for r in cur.execute(q) if any(t in r[0] for t in ("edge_mlp", "gemm_nt", "gemm_planes"))]
        for k, c, n, v in rows:
            lines.append(f"| `{k[:60]}` | {c} | {n} | {v:.6g} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
