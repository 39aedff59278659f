"""profiles/rNN_pmc_hbm.json from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) of
`VFIND_RES_LANES=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity`.
usage: pmc_hbm.py fetch_counter_collection.csv write_counter_collection.csv out.json [n_reads row_bytes]"""
import csv, json, re, sys, collections

def avg(path, counter, min_grid=60000):      # (k_log_apply: one workgroup per 8 192 slots = 65 536 threads at 2^21 slots)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or int(r.get("Grid_Size", 0)) < min_grid:
            continue
        acc[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")].append(float(r["Counter_Value"]) * 1024.0)   # KB -> bytes
    return {k: sum(v) / len(v) for k, v in acc.items()}

fetch, write = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
n = int(sys.argv[4]) if len(sys.argv) > 4 else 10_000_000
row_b = int(sys.argv[5]) if len(sys.argv) > 5 else 40
out = {
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (each with --kernel-trace only) of "
              "`VFIND_RES_LANES=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extras` (one batch lane, so no two "
              "batches' kernels overlap in the counters); per-launch averages over the full-size launches, KB as reported x 1024.",
    "gfx950_correction": "FETCH_SIZE tallies 128-B streaming requests as 64 B (MI355X_MICROARCH.md, HBM). Calibrated on k_scan, whose "
                         "only fetches are the packed rows + lengths: it reports about half the known bytes, so the streaming part of "
                         "every kernel is doubled; random 64-B line reads (count-table probes) are 64-B requests and are taken as reported.",
    "kernels": {},
}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    e = {"fetch_reported": int(f), "write_reported": int(w), "traffic": None, "note": "4-byte gathers / mixed widths: uncalibrated"}
    if "k_scan" in k:
        stream = n * (row_b + 2)                       # packed rows + 2-byte lengths
        e["traffic"] = int(f + stream / 2 + w)
        e["note"] = ("row + length stream (%d MB real, reported as half; calibrated on the first version of this kernel, whose only "
                     "fetches were that stream) + candidate-verification re-reads as reported + WRITE" % (stream // 1_000_000))
    elif "k_finish" in k:
        stream = n * (row_b + 4)                       # rows are fetched unconditionally, one 4-byte span per read
        e["traffic"] = int(f + stream / 2 + w)
        e["note"] = ("row + span stream (%d MB real, reported as half) + table-line fetches as reported + WRITE "
                     "(slot stores, CAS, the count log; no count line is touched since round 3)" % (stream // 1_000_000))
    elif "k_log" in k:
        e["traffic"] = int(f + w)
        e["note"] = "the count log applied: entries read (4- and 8-byte gathers, taken as reported) + WRITE (grouped entries / count adds)"
    out["kernels"][k] = e
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
