"""profiles/rNN_requests.json from the single-counter passes of scripts/pmc_requests.sh
usage: pmc_requests_json.py out.json summary.txt [summary2.txt ...]      (summaries: one JSON line per counter)"""
import json
import sys

rows = {}
for f in sys.argv[2:]:
    for line in open(f):
        if line.startswith("{"):
            j = json.loads(line)
            rows[j["counter"]] = {k.replace("vf::", ""): v["avg"] for k, v in j["kernels"].items()}
need = ["TCC_REQ_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_WRREQ_64B_sum",
        "TCC_EA0_ATOMIC_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCC_EA0_RDREQ_LEVEL_sum"]
missing = [c for c in need if c not in rows]
if missing:
    raise SystemExit("missing counters: " + " ".join(missing))
ks = [k for k in rows["TCC_REQ_sum"] if k.startswith("k_") and not k.startswith("k_peek")]
per_step = {k: (2 if k.startswith("k_bmfilter") or k.startswith("k_dpall") or k.startswith("k_filter") else 1) for k in ks}
out = {"method": "rocprofv3 --kernel-trace --pmc <ONE counter> per pass (scripts/pmc_requests.sh): `VFIND_RES_LANES=1 python3 bench.py --steps 3 --warmup 1 "
                 "--no-cpu-baseline --no-parity --no-extras`, one batch lane (counter collection serialises kernels anyway), per-launch averages over the "
                 "full-size launches.  Asked for TOGETHER the TCC counters are refused (error 38, 'Request exceeds the capabilities of the hardware to "
                 "collect', and an abort: what rounds 1-2 recorded as a hang); one counter per pass takes two seconds and none has hung.",
       "workload": "BASELINE configs[1]: 10 M x 150 bp reads a step, 20 bp adapters, defaults; k_bmfilter and k_dpall run twice a step (prefix, suffix pass)",
       "launches_per_step": per_step, "per_launch": {k: {c: rows[c].get(k) for c in rows} for k in ks}, "derived_per_launch": {}}
for k in ks:
    g = lambda c: rows[c].get(k, 0.0)
    rd, r128, r64 = g("TCC_EA0_RDREQ_sum"), g("TCC_EA0_RDREQ_128B_sum"), g("TCC_EA0_RDREQ_64B_sum")
    w, w64, at = g("TCC_EA0_WRREQ_sum"), g("TCC_EA0_WRREQ_64B_sum"), g("TCC_EA0_ATOMIC_sum")
    out["derived_per_launch"][k] = {
        "l2_requests": round(g("TCC_REQ_sum")), "l2_hit_frac": round(g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1), 3),
        "memory_side_read_requests": round(rd), "of_them_128B": round(r128), "memory_side_write_requests": round(w), "of_them_64B": round(w64),
        "memory_side_atomics": round(at), "read_MB": round((r128 * 128 + r64 * 64) / 1e6, 1), "write_MB": round((w64 * 64 + max(w - w64 - at, 0) * 32) / 1e6, 1),
        "avg_l2_read_latency_cycles": round(g("TCP_TCC_READ_REQ_LATENCY_sum") / max(g("TCP_TCC_READ_REQ_sum"), 1)),
        "avg_memory_read_latency_cycles": round(g("TCC_EA0_RDREQ_LEVEL_sum") / max(rd, 1)), "launches_per_step": per_step[k]}
d = out["derived_per_launch"]
tot = lambda f: sum(d[k][f] * per_step[k] for k in ks)
out["derived_per_step"] = {
    "l2_requests": round(tot("l2_requests")), "memory_side_read_MB": round(tot("read_MB")), "memory_side_write_MB": round(tot("write_MB")),
    "memory_side_requests": round(tot("memory_side_read_requests") + tot("memory_side_write_requests")),
    "note": "memory-side = the L2's fabric interface (TCC_EA0_*): Infinity-Cache hits are in these numbers (MI355X_MICROARCH.md, HBM).  EVERY read request "
            "is a 128-byte one - also the random slot-line fetches of k_finish and the 40-byte row gathers of the filter and DP kernels - so FETCH_SIZE "
            "(= requests x 64 B) halves all of them, not just the streams.  traffic of a kernel = read_MB + write_MB of its launch."}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: (v["l2_requests"], v["read_MB"], v["write_MB"]) for k, v in d.items()}))
print(out["derived_per_step"])
