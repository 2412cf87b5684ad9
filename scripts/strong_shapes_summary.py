"""gpurun_out/r6_strong_shapes.jsonl (scripts/gpu_strong_shapes.sh) -> profiles/r6_strong_shapes.json: per per-rank shape the best measured rate, and the
PREDICTED strong-scaling curve N x rate(256 / N) beside the weak one N x rate(256) -- the sampler has no data-path collective and the fine-tune step one 49.4 MB
all-reduce per 50 timesteps, so a rank of an N-GPU run does exactly the single-GPU work of its shape.
usage: python scripts/strong_shapes_summary.py <in.jsonl> <out.json>"""
import json
import sys

rows = [json.loads(ln) for ln in open(sys.argv[1]) if ln.startswith("{")]
out = {"source": "scripts/gpu_strong_shapes.sh on one MI355X", "runs": [], "sampler": {}, "fine_tune": {}}
for r in rows:
    kind, b = r["tag"].split()[0], int(r["tag"].split("B=")[1].split()[0])
    rec = {"tag": r["tag"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
           "chains_or_groups": r["config"].get("concurrent_chains", r["config"].get("concurrent_groups")), "roofline_frac": r["roofline"]["frac"]}
    out["runs"].append(rec)
    tab = out["sampler" if kind == "sample" else "fine_tune"]
    cur = tab.get(str(b))
    if cur is None or rec["value"] > cur["value"]:
        tab[str(b)] = dict(rec, automatic=("streams=0" in r["tag"] or "groups=0" in r["tag"]))
    if "streams=0" in r["tag"] or "groups=0" in r["tag"]:
        tab.setdefault("_auto", {})[str(b)] = rec["value"]
for name in ("sampler", "fine_tune"):
    tab = out[name]
    auto = tab.pop("_auto", {})
    if "256" not in auto:
        continue
    base = auto["256"]
    pred = {}
    for n, b in ((1, 256), (2, 128), (4, 64), (8, 32)):
        if str(b) in auto:
            pred[str(n)] = {"per_rank_batch": b, "rate_per_rank_automatic": auto[str(b)], "strong_total": n * auto[str(b)], "strong_speedup": n * auto[str(b)] / base,
                            "best_per_rank": tab[str(b)]["value"], "strong_speedup_best": n * tab[str(b)]["value"] / tab["256"]["value"],
                            "weak_total": n * base, "weak_speedup": float(n)}
    tab["predicted_scaling"] = pred
json.dump(out, open(sys.argv[2], "w"), indent=1)
for name in ("sampler", "fine_tune"):
    for n, p in out[name].get("predicted_scaling", {}).items():
        print(name, "N =", n, "B/rank", p["per_rank_batch"], "rate/rank %.2f" % p["rate_per_rank_automatic"], "strong x%.2f" % p["strong_speedup"], "(best x%.2f)" % p["strong_speedup_best"])
