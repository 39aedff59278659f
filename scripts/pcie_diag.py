"""where the other half of the PCIe link goes: on ONE box, one process -
  (1) raw pinned H2D of torch's pinned allocator,
  (2) raw H2D out of the library's own staging (hipHostMalloc in vf_batch_acquire), no kernels,
  (3) the pinned batch API leg (copies + whole device path), 8 slots.
usage: pcie_diag.py [trace]   (trace: only leg 3, two rounds - for rocprofv3 --kernel-trace --memory-copy-trace)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vfind_amd
from benchlib import synth
trace = len(sys.argv) > 1 and sys.argv[1] == "trace"
ads = synth.adapters(0x5EED0002, 20)
gen = synth.Generator(0x5EED0002, ads)
W = gen.row_dwords
ql, nb = 8, 1_250_000
row_b = W * 4 + 2
if not trace:
    n = 512 << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(4):
        for o in range(0, n, n // 8):
            d[o:o + n // 8].copy_(h[o:o + n // 8], non_blocking=True)
    torch.cuda.synchronize()
    print(f"(1) torch pinned H2D, 64 MiB pieces: {4 * n / (time.perf_counter() - t0) / 1e9:.1f} GB/s", flush=True)
    del h
eng = vfind_amd.Engine(ads, queue_len=ql, table_slots=1 << 21)
stag = []
for k in range(ql):
    hdl, r, l = eng.batch_acquire(nb, W)
    gen.packed_into(k * nb, nb, r.ctypes.data, l.ctypes.data, 16)
    stag.append((r.ctypes.data, r.nbytes))
    eng.batch_submit(hdl, nb)
eng.sync(); eng.clear()
if not trace:
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    dst = torch.empty(stag[0][1], dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tot = 0
    for r in range(4):
        for p, nbytes in stag:
            assert hip.hipMemcpyAsync(dst.data_ptr(), p, nbytes, 1, None) == 0
            tot += nbytes
    torch.cuda.synchronize()
    print(f"(2) library staging (hipHostMalloc) H2D, {stag[0][1] >> 20} MiB pieces, null stream: {tot / (time.perf_counter() - t0) / 1e9:.1f} GB/s", flush=True)
for _ in range(2 * ql):                                   # untimed, at full submit rate: the table grows for a whole pipeline once
    hdl, r, l = eng.batch_acquire(nb, W)
    eng.batch_submit(hdl, nb)
eng.sync()
rounds = 2 if trace else 6
t0 = time.perf_counter()
for _ in range(rounds * ql):
    hdl, r, l = eng.batch_acquire(nb, W)
    eng.batch_submit(hdl, nb)
eng.sync()
dt = time.perf_counter() - t0
reads = rounds * ql * nb
print(f"(3) batch API, queue_len={ql} batch={nb}: {reads / dt / 1e6:.1f} M reads/s, {reads * row_b / dt / 1e9:.2f} GB/s over the link", flush=True)
eng.close()
