"""does hipMemcpyAsync (page-locked host -> device) return at once?  eight 47 MB copies on eight non-blocking streams,
host time of every call, then the same with a small kernel-free event record behind each"""
import ctypes as C, time, sys
import torch
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
torch.cuda.init(); torch.zeros(1, device="cuda")
nbytes = 47 << 20
hs, ds, st = [], [], []
for k in range(8):
    p = C.c_void_p(); assert hip.hipHostMalloc(C.byref(p), nbytes, 0) == 0; hs.append(p)
    ds.append(torch.empty(nbytes, dtype=torch.uint8, device="cuda"))
    s = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0; st.append(s)
torch.cuda.synchronize()
for rnd in range(3):
    t = []
    t0 = time.perf_counter()
    for k in range(8):
        a = time.perf_counter()
        assert hip.hipMemcpyAsync(ds[k].data_ptr(), hs[k], nbytes, 1, st[k]) == 0
        t.append((time.perf_counter() - a) * 1e3)
    for k in range(8): hip.hipStreamSynchronize(st[k])
    dt = time.perf_counter() - t0
    print("call ms:", " ".join(f"{x:.2f}" for x in t), f"| {8 * nbytes / dt / 1e9:.1f} GB/s")
