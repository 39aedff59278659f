#!/usr/bin/env python3
"""bench.py -- reads/s of the variant-extraction hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the whole device path (exact scan -> semi-global DP fallback -> extract ->
translate -> hash-count, table cleared first) over one batch of synthetic reads resident in HBM.
Workload at N=1 = BASELINE.json configs[1]: 10 M synthetic 150 bp reads, 20 bp adapters, default
alignment parameters (SURVEY.md 8d read model: 0.5 % substitutions / 0.05 % indels in the adapters,
5 % adapter-free reads, Zipf(1.1) variants from a 10^6 library).  With N GPUs every rank runs the
same per-GPU workload on its own read range (weak scaling, no data-path collective); the final
(key,count) table merge over RCCL is timed separately and reported as merge_ms.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def b_alg(read_len: int) -> int:
    """algorithmic bytes per read: packed read in + one 4-byte (start,end) out"""
    return (2 * read_len + 7) // 8 + 4


def tie_sensitive_fraction(data, offs, ads, oracle, budget_s=8.0, max_reads=400):
    """Among the probe's reads that needed an alignment and were accepted for that adapter: the fraction whose optimal
    alignments do not all have the same `length` (so parasail's tie-breaking, which nothing pins, decides the cut)."""
    from oracle import pyref
    st, en = oracle.spans((data, offs), ads, n_threads=8)
    t0 = time.perf_counter()
    seen = tied = 0
    raw = data.tobytes()
    for i in range(len(offs) - 1):
        if seen >= max_reads or time.perf_counter() - t0 > budget_s:
            break
        r = raw[int(offs[i]):int(offs[i + 1])]
        for ad, coord in ((ads[0], st[i]), (ads[1], en[i])):
            if coord < 0 or oracle.find_exact(r, ad) >= 0:
                continue                                   # not accepted, or found exactly: no alignment involved
            _, ends = pyref.cooptimal(ad.encode(), r)
            seen += 1
            tied += len({e[2] for e in ends}) > 1
    return {"value": round(tied / seen, 4) if seen else None, "alignments_examined": seen, "tied": tied,
            "what": "accepted alignments of the benchmark input whose co-optimal paths differ in length (oracle/pyref.py enumerator); "
                    "kernel == oracle there under tie policy P, oracle == parasail is unpinned"}


def other_config_leg(vfind_amd, synth, oracle, name, device, host_threads, n, n_parity, L, adapter_len, variants, sub_rate, indel_rate, seed,
                     steps=10, warmup=2):
    """one of BASELINE.json's other shapes on this GPU, resident in HBM like the headline: its own ms_per_step and path roofline
    over n reads a step, and the table of the first n_parity of those reads compared - whole - with the CPU oracle's"""
    import numpy as np
    import torch
    ads = synth.adapters(seed, adapter_len)
    gen = synth.Generator(seed, ads, read_len=L, n_variants=variants, sub_rate=sub_rate, indel_rate=indel_rate)
    rows_h, lens_h = gen.packed(0, n, host_threads)
    dev = torch.device("cuda", device)
    rows = torch.from_numpy(rows_h).to(dev)
    lens = torch.from_numpy(lens_h.view(np.int16)).to(dev)
    del rows_h, lens_h
    torch.cuda.synchronize(dev)
    slots = 1 << 16
    while slots < max(2 * min(variants, n), n // 8):
        slots <<= 1
    eng = vfind_amd.Engine(ads, device=device, table_slots=slots, arena_bytes=1 << 30)
    try:
        def step():
            eng.clear()
            eng.process_device(rows.data_ptr(), gen.row_dwords, lens.data_ptr(), n)
        for _ in range(warmup):
            step()
        eng.sync()
        eng.reset_stats()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        eng.sync()
        dt = time.perf_counter() - t0
        st = eng.stats()
        # the same steps into the table as the last one left it: every key is there already (a run many times longer than its
        # variant library - BASELINE configs[4]: 10^9 reads over 10^7 variants - spends nearly all of its time in this state)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.process_device(rows.data_ptr(), gen.row_dwords, lens.data_ptr(), n)
        eng.sync()
        dt_kept = time.perf_counter() - t0
        eng.clear()
        eng.process_device(rows.data_ptr(), gen.row_dwords, lens.data_ptr(), n_parity)
        got = eng.finish_dict()
        dup = int(eng.stats()["dup_rows"])
        data, offs = gen.ascii(0, n_parity, host_threads)
        tq = time.perf_counter()
        want = oracle.count((data, offs), ads, n_threads=host_threads)
        cpu_dt = time.perf_counter() - tq
        if got != want:
            raise SystemExit(f"PARITY FAILED ({name}): device table differs from the CPU oracle")
        ms = dt / steps * 1e3
        gbps = n * b_alg(L) / (ms * 1e-3) / 1e9
        return {"workload": f"{n} synthetic {L} bp reads resident in HBM, {adapter_len} bp adapters, {variants} variants, "
                            f"substitutions {sub_rate}, indels {indel_rate}, default alignment params",
                "value": round(n * steps / dt / 1e6, 2), "unit": "M reads/s", "ms_per_step": round(ms, 4), "steps": steps,
                "roofline_path": {"bound": "hbm", "achieved": round(gbps, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": round(gbps / HBM_PEAK_GBPS, 6), "alg_bytes_per_read": b_alg(L)},
                "alignments_per_read": round((st["dp_prefix"] + st["dp_suffix"]) / steps / n, 3),
                "table_kept": {"value": round(n * steps / dt_kept / 1e6, 2), "unit": "M reads/s", "ms_per_step": round(dt_kept / steps * 1e3, 4),
                               "what": "the same steps without the clear in front: every key already in the table (`value` clears the table every step)"},
                "parity_full_table": True, "parity_reads": n_parity, "table_rows": len(got), "dup_rows": dup,
                "cpu_oracle_M_reads_per_s": round(n_parity / cpu_dt / 1e6, 3), "cpu_threads": host_threads}
    finally:
        eng.close()
        del rows, lens


def h2d_raw_gbps(hip_host_ptrs, nbytes, device):
    """raw rate of the link out of the library's own page-locked staging, same process, same box: the batches' buffers copied
    to a scratch device buffer with nothing else going on (what `link_GBps` of the leg is read against)"""
    import ctypes as C
    import torch
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    dst = torch.empty(nbytes, dtype=torch.uint8, device=torch.device("cuda", device))
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    tot = 0
    for _ in range(3):
        for p in hip_host_ptrs:
            if hip.hipMemcpyAsync(dst.data_ptr(), p, nbytes, 1, None) != 0:
                return None
            tot += nbytes
    torch.cuda.synchronize(device)
    return round(tot / (time.perf_counter() - t0) / 1e9, 2)


def pcie_inclusive_leg(vfind_amd, gen, ads, n, L, device, slots, host_threads, rounds=6, queue_len=8):
    """reads/s with the batches in page-locked HOST memory when the clock starts: vf_batch_acquire / vf_batch_submit, queue_len
    slots, hipMemcpyAsync of batch k + 1 overlapped with the kernels of batch k.  The staging is filled once (the host decoder's
    own speed is not part of this leg) and re-submitted `rounds` times."""
    nb = n // queue_len
    eng = vfind_amd.Engine(ads, device=device, table_slots=slots, arena_bytes=1 << 30, queue_len=queue_len)
    try:
        W = gen.row_dwords
        stag = []
        for k in range(queue_len):                       # fill every slot's staging once
            h, r, l = eng.batch_acquire(nb, W)
            gen.packed_into(k * nb, nb, r.ctypes.data, l.ctypes.data, host_threads)
            stag.append(r.ctypes.data)
            eng.batch_submit(h, nb)
        eng.sync()
        raw = h2d_raw_gbps(stag, nb * W * 4, device)
        # two untimed rounds at full submit rate: with every slot in flight the table grows once to the size that holds a
        # whole pipeline of possibly-new keys (a one-off rehash that a long job does not notice and a 50 ms window would)
        for _ in range(2 * queue_len):
            h, r, l = eng.batch_acquire(nb, W)
            eng.batch_submit(h, nb)
        eng.sync()
        eng.clear()
        t0 = time.perf_counter()
        for _ in range(rounds * queue_len):
            h, r, l = eng.batch_acquire(nb, W)           # the oldest slot: its rows are still in the staging
            eng.batch_submit(h, nb)
        eng.sync()
        dt = time.perf_counter() - t0
        reads = rounds * queue_len * nb
        row_b = W * 4 + 2
        link = reads * row_b / dt / 1e9
        return {"value": round(reads / dt / 1e6, 2), "unit": "M reads/s", "link_GBps": round(link, 2),
                "link_peak_GBps": 63.0, "h2d_raw_GBps": raw, "frac_of_raw": None if not raw else round(link / raw, 3),
                "batches": rounds * queue_len, "reads_per_batch": nb, "queue_len": queue_len,
                "what": "packed reads in the library's page-locked staging -> hipMemcpyAsync -> whole device path, table not cleared between batches"}
    finally:
        eng.close()


def end_to_end_leg(vfind_amd, oracle, gen, ads, n_reads, device, slots, host_threads, sample_reads=2_000_000):
    """BASELINE configs[2] in shape - "2 x gzipped FASTQ, host decode / H2D overlap stressed" - as a find_variants caller gets it:
    TWO multi-member gzip FASTQ files of the bench's own read model (R1 and R2 as two independent inputs: the reference has no
    paired mode, lib.rs:220 takes one path) run into ONE context, then the frame.  Timed with the reference's default
    n_threads = 3 and with all host threads of the box's share; the stage timers of vf_stats (host steady_clock) say where the
    wall time goes; the CPU restatement (oracle.count_fastq: the reference's own shape, one reader + W workers + one reducer)
    is timed beside it on a bounded sample file of the same stream, whose table the device must reproduce."""
    import tempfile
    import zlib
    from benchlib import fastq
    half = n_reads // 2
    members = max(8, half // 160_000)                      # ~ 50 MB of text a member
    out = {"reads": 2 * half, "files": 2, "members_per_file": members}
    with tempfile.TemporaryDirectory() as d:
        paths = [os.path.join(d, "bench_R1.fq.gz"), os.path.join(d, "bench_R2.fq.gz")]
        sample = os.path.join(d, "sample.fq.gz")
        t0 = time.perf_counter()
        size = sum(fastq.write_fastq_gz(p, gen, half, members=members, threads=host_threads, first=k * half) for k, p in enumerate(paths))
        fastq.write_fastq_gz(sample, gen, sample_reads, members=16, threads=host_threads)
        out["file_MB"] = round(size / 1e6, 1)
        out["write_s"] = round(time.perf_counter() - t0, 1)
        # zlib 1.2.11 (through Python's binding) on the first member of the same file, one thread: what the library's own decoder
        # (vf_inflate.cpp) is read against
        raw = open(paths[0], "rb").read(1 << 28)
        tz = time.perf_counter()
        dz = zlib.decompressobj(31)
        text = dz.decompress(raw)
        out["zlib_MBps_of_text_per_thread"] = round(len(text) / (time.perf_counter() - tz) / 1e6, 1)
        del raw, text, dz
        sample_want = None
        for nt in (3, host_threads):
            with vfind_amd.Engine(ads, device=device, table_slots=slots, arena_bytes=1 << 30, n_threads=nt, queue_len=4) as e:
                e.run_fastq(sample)                        # first call allocates staging and workspaces; its table is the parity sample
                got = e.finish_dict()
                if sample_want is None:
                    tq = time.perf_counter()
                    sample_want = oracle.count_fastq(sample, ads, n_threads=host_threads)
                    t_all = time.perf_counter() - tq
                    tq = time.perf_counter()
                    oracle.count_fastq(sample, ads, n_threads=3)
                    t_3 = time.perf_counter() - tq
                    out["cpu_baseline"] = {"kind": "port", "unit": "M reads/s", "sample": f"a {sample_reads}-read file of the same stream (16 members)",
                                           "workers_3": round(sample_reads / t_3 / 1e6, 3), f"workers_{host_threads}": round(sample_reads / t_all / 1e6, 3),
                                           "what": "oracle.count_fastq: zlib reader thread -> W workers -> one serial reducer (the reference's shape, lib.rs:233-234, 271-306)"}
                if got != sample_want:
                    raise SystemExit("PARITY FAILED (end to end): vf_run_fastq table differs from oracle.count_fastq on the sample file")
                e.clear()
                e.reset_stats()
                t0 = time.perf_counter()
                for p in paths:
                    e.run_fastq(p)
                t_run = time.perf_counter() - t0
                frame = e.finish_arrays()
                dt = time.perf_counter() - t0
                st = e.stats()
                thr = max(int(st["fq_inflate_threads"]), 1)
                leg = {"value": round(2 * half / dt / 1e6, 2), "unit": "M reads/s", "seconds": round(dt, 3), "rows": len(frame[2]), "dup_rows": int(st["dup_rows"]),
                       "stages_ms": {"wall_run_fastq": round(st["fq_wall_ms"], 1), "finish": round((dt - t_run) * 1e3, 1),
                                     "inflate_thread_sum": round(st["fq_inflate_ms"], 1), "inflate_per_thread": round(st["fq_inflate_ms"] / thr, 1),
                                     "sequencer_waits_for_text": round(st["fq_seq_wait_ms"], 1), "sequencer_waits_for_queue": round(st["fq_queue_full_ms"], 1),
                                     "submit_thread_sum": round(st["fq_submit_ms"], 1), "of_it_waiting_for_a_slot": round(st["fq_slot_wait_ms"], 1),
                                     "of_it_waiting_for_the_line_index": round(st["fq_index_wait_ms"], 1), "final_drain": round(st["fq_drain_ms"], 1)},
                       "inflate_threads": thr, "chunks": int(st["fq_chunks"]),
                       "text_MB": round(st["fq_text_bytes"] / 1e6, 1),
                       "inflate_MBps_of_text_per_thread": round(st["fq_text_bytes"] / max(st["fq_inflate_ms"], 1e-3) / 1e3, 1),
                       "text_GBps_overall": round(st["fq_text_bytes"] / dt / 1e9, 2)}
                # device time of the same path per stage: the sample file again with every launch bracketed (not in `value`)
                e.clear()
                e.reset_stats()
                e.set_profiling(1)
                e.run_fastq(sample)
                e.set_profiling(0)
                sd = e.stats()
                leg["device_ms_per_M_reads"] = {k: round(sd[f] / (sample_reads / 1e6), 3) for k, f in (
                    ("h2d_text", "ms_h2d"), ("line_index_frame_pack", "ms_frame"), ("scan", "ms_scan"), ("filter", "ms_filter"), ("dp", "ms_dp_window"),
                    ("finish", "ms_finish"), ("count_log", "ms_count_log"), ("generic", "ms_generic"))}
                out[f"n_threads_{nt}"] = leg
    out["parity_sample_file"] = True
    out["what"] = ("vf_run_fastq x 2 files + vf_finish on gzip FASTQ of the same read model (never `value`); stages: host steady_clock timers of vf_stats "
                   "(thread sums where several threads work), device_ms: HIP events around every launch in a separate pass over the sample file")
    return out


def valu_util_from_profiles():
    """per-kernel VALU utilisation from the SQ counter pass committed under profiles/ (scripts/pmc_sq.sh)"""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_sq.json")))
        return {"source": "profiles/r04_pmc_sq.json", "definition": j.get("definition"),
                "kernels": {k: v.get("valu_util") for k, v in j.get("kernels", {}).items()}}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)     # (the first dozen steps of a process run ~1 % slower than the rest)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--adapter-len", type=int, default=20)
    ap.add_argument("--variants", type=int, default=1_000_000)
    ap.add_argument("--sub-rate", type=float, default=0.005)
    ap.add_argument("--indel-rate", type=float, default=0.0005)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED0002)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU work budget (core-seconds) of the baseline sample")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--table-slots", type=int, default=0, help="count-table slots (0 = 2x the variant library, power of two)")
    ap.add_argument("--no-extras", action="store_true", help="skip the legs beside the headline: result build, PCIe-inclusive, end-to-end FASTQ, tie-sensitive fraction")
    ap.add_argument("--e2e-reads", type=int, default=20_000_000, help="reads of the two gzip FASTQ files of the end-to-end leg (BASELINE configs[2] in shape)")
    ap.add_argument("--other-config-reads", type=int, default=8_000_000, help="resident reads per step of each of the other BASELINE shapes (extras)")
    ap.add_argument("--other-config-parity-reads", type=int, default=4_000_000, help="reads of those shapes whose whole table is compared with the CPU oracle")
    ap.add_argument("--merge-transport", default="auto", choices=["auto", "rccl", "torch"],
                    help="final table reduce: rccl = the library's own exchange (vf_comm_init / vf_table_allreduce), torch = the same "
                         "algorithm with torch.distributed moving the bytes; auto = rccl when the backend is nccl")
    ap.add_argument("--lib", default=None, help=argparse.SUPPRESS)      # diagnosis: another build of the library (make ablation)
    ap.add_argument("--dry-run", action="store_true", help="launcher / rank plumbing only (no device work): what the CPU test of --gpus N runs")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the same
        # torch.distributed.run line the driver uses) as a CHILD process, before anything here has touched the GPU,
        # and leave with its exit code.
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or run without a launcher)")

    import numpy as np
    import torch
    import torch.distributed as dist

    if args.dry_run:
        # the N>1 plumbing without a device: ranks rendezvous, barrier, reduce a clock the way the timed region does
        if world > 1:
            dist.init_process_group("gloo" if args.backend != "nccl" or not torch.cuda.is_available() else "nccl")
            dist.barrier()
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.barrier()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "max_over_ranks": float(t.item()),
                              "scaling": "weak"}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    import vfind_amd
    from benchlib import synth
    if args.lib:
        from vfind_amd import _lib as _vl
        _vl.SO_PATH = os.path.abspath(args.lib)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    wire = dev if args.backend == "nccl" else torch.device("cpu")

    def barrier():
        if world > 1:
            if args.backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()

    n = args.reads
    L = args.read_len
    host_threads = 16 if world == 1 else 8
    ads = synth.adapters(args.seed, args.adapter_len)
    gen = synth.Generator(args.seed, ads, read_len=L, n_variants=args.variants, sub_rate=args.sub_rate,
                          indel_rate=args.indel_rate)
    t0 = time.perf_counter()
    rows_h, lens_h = gen.packed(rank * n, n, host_threads)
    t_gen = time.perf_counter() - t0
    rows = torch.from_numpy(rows_h).to(dev)
    lens = torch.from_numpy(lens_h.view(np.int16)).to(dev)
    del rows_h, lens_h
    torch.cuda.synchronize(dev)

    slots = args.table_slots or (1 << 16)
    # load factor <= 0.5 at worst: the library, plus room for the keys that differ only by an alignment-derived cut
    while not args.table_slots and slots < max(2 * min(args.variants, n), n // 8):
        slots <<= 1
    eng = vfind_amd.Engine(ads, device=local_rank, table_slots=slots, arena_bytes=1 << 30)

    # ---- parity gate (rank 0): device path == CPU oracle on a sample of the same reads -----------
    parity = None
    cpu = None
    tie = None
    oracle_full, oracle_full_n = None, 0
    if world > 1:
        args.no_cpu_baseline = True       # the CPU baseline is timed at N=1 only (the other ranks would wait for it)
    if rank == 0 and not (args.no_parity and args.no_cpu_baseline):
        from oracle import oracle          # the checker / CPU baseline; never part of the timed GPU path
        cores = min(os.cpu_count() or 1, 16)          # a 1-GPU box gives this job a 16-core share
        probe_n = min(n, 50_000)
        pb, po = gen.ascii(0, probe_n, host_threads)
        tq = time.perf_counter()
        want = oracle.count((pb, po), ads, n_threads=cores)
        probe_dt = max(time.perf_counter() - tq, 1e-4)
        if not args.no_parity:
            eng.submit((pb, po))
            got = eng.finish_dict()
            parity = got == want
            if not parity:
                raise SystemExit("PARITY FAILED: device (sequence,count) table differs from the CPU oracle")
            eng.clear()
        if not args.no_cpu_baseline:
            rate = probe_n / probe_dt                                     # reads/s on all cores
            sample = int(min(n, max(probe_n, rate * args.cpu_seconds / 1.0)))
            sb, so = gen.ascii(0, sample, host_threads)
            best = None
            for _ in range(2):
                tq = time.perf_counter()
                oracle_full = oracle.count((sb, so), ads, n_threads=cores)
                dt = time.perf_counter() - tq
                best = dt if best is None else min(best, dt)
            oracle_full_n = sample
            cpu = {"value": round(sample / best / 1e6, 4), "unit": "M reads/s", "cores": cores, "kind": "port",
                   "sample": f"first {sample} reads of the same workload, all {cores} host threads, each with its own count table merged at "
                             "the end (BASELINE.md 3 planned one serial reducer: this shape favours the CPU), best of 2 runs; "
                             "CPU restatement of vfind 0.3.1 (reference not buildable: no rustc/parasail)"}
            # the reference's default worker count (n_threads = 3, lib.rs:178) on a tenth of that sample
            s3 = max(probe_n, sample // 10)
            tq = time.perf_counter()
            oracle.count((sb[: s3 * L], so[: s3 + 1]), ads, n_threads=3)
            cpu["at_reference_default_threads"] = {"value": round(s3 / max(time.perf_counter() - tq, 1e-4) / 1e6, 4), "cores": 3, "sample_reads": s3}
            # BASELINE.md 3's shape: the reference's own threading (seq_io::parallel_fastq, lib.rs:271-306) - W workers run the
            # `work` closure on record sets, ONE thread reduces every record in input order; best of 3, at the reference's
            # default W = 3 and at every core of this job's share
            ref_shape = {}
            for w in (3, cores):
                best_w = None
                for _ in range(3):
                    tq = time.perf_counter()
                    oracle.count_reference_shape((sb[: s3 * L], so[: s3 + 1]), ads, n_workers=w)
                    dtw = time.perf_counter() - tq
                    best_w = dtw if best_w is None else min(best_w, dtw)
                ref_shape[f"workers_{w}"] = {"value": round(s3 / max(best_w, 1e-4) / 1e6, 4), "unit": "M reads/s", "workers": w, "reducers": 1}
            cpu["reference_shape"] = dict(ref_shape, sample_reads=s3, runs="best of 3",
                                          what="W worker threads -> one serial reducer on the calling thread (lib.rs:271-306)")
            del sb, so
        # fraction of this input's accepted alignments whose `length` depends on the aligner's tie-breaking (BASELINE.md 4,
        # SURVEY App. B.4): sampled with the exhaustive co-optimal enumerator of oracle/pyref.py on reads of the probe
        if not args.no_extras and not args.no_parity:
            tie = tie_sensitive_fraction(pb, po, ads, oracle, budget_s=8.0)
        del pb, po
    eng.reset_stats()

    keep_table = os.environ.get("VFIND_BENCH_KEEP_TABLE") == "1"      # experiment only: steps then count into a warm table

    def step():
        if not keep_table:
            eng.clear()
        eng.process_device(rows.data_ptr(), gen.row_dwords, lens.data_ptr(), n)

    for _ in range(args.warmup):
        step()
    eng.sync()
    eng.reset_stats()
    # HIP events around the dominant kernel (k_finish) only inside the timed region: a bracket is two event packets
    # on the stream and bracketing all ~20 launches of a step costs 6 % of it; the other kernels' durations come
    # from a pass of the same steps right after the timed region (VFIND_BENCH_EVENTS=1: everything in the timed region)
    ev_mode = {"0": 0, "1": 1, "2": 2}.get(os.environ.get("VFIND_BENCH_EVENTS", "3"), 3)
    eng.set_profiling(ev_mode)
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.sync()
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    eng.set_profiling(False)
    st = eng.stats()
    st_k, k_steps = st, max(args.steps, 1)
    # the same window four more times (N = 1; `value`, `ms_per_step` and `steps` stay those of the FIRST window): boxes of the
    # pool differ by up to 10 %, and one 20-step window is a thin basis for a 3 % conclusion
    spread = None
    if world == 1 and not args.no_extras:
        wins = [elapsed / args.steps * 1e3]
        for _ in range(4):
            torch.cuda.synchronize(dev)
            tw = time.perf_counter()
            for _ in range(args.steps):
                step()
            eng.sync()
            wins.append((time.perf_counter() - tw) / args.steps * 1e3)
        sw = sorted(wins)
        spread = {"windows": len(wins), "steps_per_window": args.steps, "ms_per_step": [round(x, 4) for x in wins],
                  "min": round(sw[0], 4), "median": round(sw[len(sw) // 2], 4), "max": round(sw[-1], 4),
                  "M_reads_per_s_median": round(n / (sw[len(sw) // 2] * 1e-3) / 1e6, 1)}

    # ---- full-size parity: the table left by the last timed step (resident packed path, all n reads)
    # against the oracle's table of the same reads, when the CPU baseline sample covered them all
    parity_full = None
    dup_rows_full = None
    if rank == 0 and oracle_full is not None and oracle_full_n == n and not args.no_parity and not keep_table:
        parity_full = eng.finish_dict() == oracle_full
        if not parity_full:
            raise SystemExit("PARITY FAILED at full size: device table differs from the CPU oracle")
        dup_rows_full = int(eng.stats()["dup_rows"])
    oracle_full = None

    # ---- beside the headline (rank 0 at N = 1; none of it is `value`) ---------------------------------------------
    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:
        # the (sequence, count) frame of the last step's table: export, device dedupe + sort, one copy back
        keys, offs, cnts = eng.finish_arrays()
        fs = eng.stats()
        extras["finish"] = {"ms": round(fs["ms_result_build"], 3), "rows": int(fs["result_rows"]), "key_bytes": int(offs[-1]) if len(offs) else 0,
                            "what": "vf_finish: table export, strings expanded + duplicates summed + rows sorted (count desc, sequence asc) on the device, one copy back"}
        del keys, offs, cnts
        extras["pcie_inclusive"] = pcie_inclusive_leg(vfind_amd, gen, ads, n, L, local_rank, slots, host_threads)
        from oracle import oracle as _oracle_e2e            # checker + CPU baseline of the leg; outside every timed device region
        extras["end_to_end"] = end_to_end_leg(vfind_amd, _oracle_e2e, gen, ads, args.e2e_reads, local_rank, slots, host_threads)
        if not args.no_parity and not args.no_cpu_baseline:
            # BASELINE.json configs[3] and configs[4] in shape (their read counts are 8-GPU jobs), each with its whole table checked
            from oracle import oracle as _oracle
            oc_n, oc_p = args.other_config_reads, min(args.other_config_reads, args.other_config_parity_reads)
            extras["other_configs"] = {
                "cfg4_shape_dp_on_every_read": other_config_leg(vfind_amd, synth, _oracle, "cfg 4 shape", local_rank, host_threads, oc_n, oc_p, 150, 20,
                                                                1_000_000, 0.15, 0.0, 0x5EED0004),
                "cfg5_shape_300bp_30bp_adapters": other_config_leg(vfind_amd, synth, _oracle, "cfg 5 shape", local_rank, host_threads, oc_n, oc_p, 300, 30,
                                                                   10_000_000, 0.005, 0.0005, 0x5EED0005),
            }

    # per-kernel durations of every kernel kind: the same steps again with all brackets on (not in `value`)
    if ev_mode in (2, 3):
        eng.reset_stats()
        eng.set_profiling(1)
        k_steps = max(min(args.steps, 10), 1)
        for _ in range(k_steps):
            step()
        eng.sync()
        eng.set_profiling(False)
        st_k = eng.stats()

    # the metric first: max over ranks of the timed region
    counted_local = st["counted"] // max(args.steps, 1)
    if world > 1:
        t_el = torch.tensor([elapsed], dtype=torch.float64, device=wire)
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        elapsed = float(t_el.item())

    # ---- final (key,count) merge across GPUs over RCCL (xGMI), timed separately ------------------
    # Runs under a watchdog: the bench line must come out even if the exchange stalls on some fabric.
    merge_ms, merge_error = None, None
    if world > 1:
        import threading
        box = {}

        def do_merge():
            try:
                tot = torch.tensor([counted_local], dtype=torch.int64, device=wire)
                dist.all_reduce(tot)
                barrier()
                tm = time.perf_counter()
                eng.allreduce_table(dst=0, transport=args.merge_transport)
                torch.cuda.synchronize(dev)
                barrier()
                box["ms"] = (time.perf_counter() - tm) * 1e3
                if rank == 0:
                    keys, offs, cnts = eng.finish_arrays()
                    if int(cnts.sum()) != int(tot.item()):
                        box["err"] = "merged counts do not sum to the per-rank totals"
            except Exception as e:                        # noqa: BLE001 - reported in the JSON line
                box["err"] = f"{type(e).__name__}: {e}"

        th = threading.Thread(target=do_merge, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("VFIND_BENCH_MERGE_TIMEOUT", "120")))
        if th.is_alive():
            merge_error = "table merge did not finish in time"
        else:
            merge_ms, merge_error = box.get("ms"), box.get("err")
        if merge_error and "do not sum" in merge_error:
            raise SystemExit("MERGE CHECK FAILED: " + merge_error)

    if rank == 0:
        is_cfg1 = (n, L, args.adapter_len, args.variants, args.sub_rate, args.indel_rate) == (10_000_000, 150, 20, 1_000_000, 0.005, 0.0005)
        ms_per_step = elapsed / args.steps * 1e3
        value = world * n * args.steps / elapsed / 1e6
        # per kernel: (summed event ms, launches, algorithmic bytes per launch).  A DP task reads its packed
        # row (+2-byte length, 4-byte task index, 4-byte window) and writes one 4-byte span; the filter
        # and the two DP groups run once per adapter, so their per-launch task count is the mean of both.
        row_b = (2 * L + 7) // 8
        steps = max(args.steps, 1)
        tasks = (st["dp_prefix"] + st["dp_suffix"]) / steps / 2
        kern = {        # (event ms summed, launches, algorithmic bytes per launch, steps the sum covers)
            "scan": ((st["ms_scan"], st["launches_scan"], n * b_alg(L), st["launches_scan"]) if ev_mode in (2, 3) and st["launches_scan"] else
                     (st_k["ms_scan"], st_k["launches_scan"], n * b_alg(L), k_steps)),                                 # timed region (mode 2)
            "filter": (st_k["ms_filter"], st_k["launches_filter"], tasks * (row_b + 10), k_steps),
            "dp_window": (st_k["ms_dp_window"], st_k["launches_dp_window"], tasks * (row_b + 14), k_steps),
            "dp_rest": (st_k["ms_dp_rest"], st_k["launches_dp_rest"], tasks * (row_b + 14), k_steps),
            # timed region; the 4th element = the steps the bracketed launches stand for (mode 3 brackets every 3rd step's)
            "finish": (st["ms_finish"], st["launches_finish"], n * 4 + st["accepted"] / steps * (row_b + 2), max(st["launches_finish"], 1) if ev_mode == 3 else steps),
            # the batch's count log applied to the table (k_log_scatter + k_log_apply, or k_log_direct): one bracket around both
            "count_log": (st_k.get("ms_count_log", 0.0), st_k.get("launches_count_log", 0), st["counted"] / steps * 4, k_steps),
        }
        # dominant kernel = most device time among the kinds that are one kernel per launch (dp_rest brackets the
        # second window class and the side stream's group of kernels, not a single kernel)
        per_step = lambda k: kern[k][0] / kern[k][3]
        dom = max((k for k in kern if k not in ("dp_rest", "count_log")), key=per_step)
        # k_scan and k_finish run within a per cent of each other at this workload: the line names k_finish (the kernel the
        # profiles under profiles/ are read against) unless another kernel is clearly ahead of it, instead of flipping run to run
        if dom != "finish" and per_step("finish") > 0 and per_step(dom) < 1.05 * per_step("finish"):
            dom = "finish"
        ms_tot, launches, bytes_per_launch, _ = kern[dom]
        avg_ms = ms_tot / max(launches, 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # memory-side bytes of the dominant kernel per launch, from the single-counter PMC passes committed under profiles/
        # (scripts/pmc_requests.sh -> profiles/r04_requests.json: TCC_EA0 read requests x their size - all of them 128 B on this
        # chip, which is what the guide's "FETCH_SIZE halves wide reads" correction amounts to - plus write requests x 64 B)
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r04_requests.json")))["derived_per_launch"]
            prefix = {"scan": "k_scan<10", "finish": "k_finish<10"}.get(dom)
            if prefix and L == 150 and n == 10_000_000:
                traffic = next((int((v["read_MB"] + v["write_MB"]) * 1e6) for k, v in pmc.items() if k.startswith(prefix)), None)
        except Exception:
            traffic = None
        # whole path: wall time per step (two batch lanes overlap, so per-kernel times sum to more than this)
        path_gbps = n * b_alg(L) / (ms_per_step * 1e-3) / 1e9 if ms_per_step > 0 else 0.0
        out = {
            "metric": f"M reads/sec ({L} bp, {args.adapter_len} bp adapters)", "value": round(value, 3), "unit": "M reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"{n} synthetic {L} bp reads per GPU per step, {args.adapter_len} bp adapters, default "
                                   "alignment params (" + ("BASELINE.json configs[1]" if is_cfg1 else "not configs[1]: see the other keys") + "), reads resident in HBM",
                       "reads_per_gpu": n, "read_len": L, "adapter_len": args.adapter_len, "variants": args.variants,
                       "sub_rate": args.sub_rate, "indel_rate": args.indel_rate, "parallelism": f"read-sharded x{world}"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic,
                         "avg_launch_ms": round(avg_ms, 4), "alg_bytes_per_launch": int(bytes_per_launch),
                         "events": ("inside the timed region" + (", on every 3rd step's launch (the brackets cost the window ~1 %: EXPERIMENTS R4.10)" if ev_mode == 3 else "")) if (dom in ("finish", "scan") or ev_mode == 1) else "pass after the timed region",
                         "launches_timed": int(launches)},
            "roofline_path": {"bound": "hbm", "achieved": round(path_gbps, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": round(path_gbps / HBM_PEAK_GBPS, 6), "alg_bytes_per_read": b_alg(L),
                              "kernel_ms_per_step": {k: round(v[0] / v[3], 4) for k, v in kern.items()},
                              "kernel_ms_source": ("scan, finish: HIP events inside the timed region" + (" (every 3rd step's launches)" if ev_mode == 3 else "") + "; others: the same steps again right after it, every launch bracketed") if ev_mode in (2, 3) else "HIP events inside the timed region"},
            # per kernel against the same HBM peak, and the ceilings that bind before HBM does: the
            # integer VALU if every read needed both alignments (2 L m cells at the measured 11.45 instructions per
            # cell, 256 CUs x 4 SIMDs x one wave64 instruction per ~2.3 cycles at 2.1 GHz), and PCIe for streamed input
            "roofline_kernels": {k: {"achieved": round(v[2] / (v[0] / max(v[1], 1) * 1e-3) / 1e9, 1) if v[0] > 0 else 0.0,
                                     "frac": round(v[2] / (v[0] / max(v[1], 1) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5) if v[0] > 0 else 0.0,
                                     "unit": "GB/s", "avg_launch_ms": round(v[0] / max(v[1], 1), 4)} for k, v in kern.items()},
            "ceilings": {"hbm_M_reads_per_s": round(HBM_PEAK_GBPS * 1e9 / b_alg(L) / 1e6, 1),
                         "valu_all_reads_aligned_M_reads_per_s": round(1024 * 2.1e9 / 2.3 * 64 / (2 * L * args.adapter_len * 11.45) / 1e6, 1),
                         "pcie_gen5_x16_M_reads_per_s": round(63e9 / ((2 * L + 7) // 8 + 2) / 1e6, 1)},
            "cpu_baseline": cpu,
            "parity_gate": parity, "parity_full_size": parity_full, "dup_rows_full_size": dup_rows_full, "table_cleared_each_step": not keep_table,
            "merge_ms": None if merge_ms is None else round(merge_ms, 3), "merge_error": merge_error,
            "merge_transport": None if world == 1 else {"rccl": "rccl (vf_table_allreduce)"}.get(getattr(eng, "last_merge_transport", args.merge_transport), getattr(eng, "last_merge_transport", args.merge_transport)),
            "value_spread": spread, "other_configs": extras.get("other_configs"),
            "finish": extras.get("finish"), "pcie_inclusive": extras.get("pcie_inclusive"), "end_to_end": extras.get("end_to_end"),
            "tie_sensitive_frac": tie, "valu_util": valu_util_from_profiles(),
            "per_step": {"dp_prefix_tasks": st["dp_prefix"] // steps, "dp_suffix_tasks": st["dp_suffix"] // steps, "insert_path_reads": st["insert_path"] // steps,
                         "accepted": st["accepted"] // steps, "counted": st["counted"] // steps,
                         "table_entries": st["table_entries"]},
            "gen_s": round(t_gen, 2),
        }
        print(json.dumps(out), flush=True)
    if merge_error == "table merge did not finish in time":
        sys.stdout.flush()
        sys.stderr.write("bench.py: the table merge did not finish in time (stuck collective)\n")
        sys.stderr.flush()
        os._exit(3)                                       # the stuck collective would block close() as well; never rc 0
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
