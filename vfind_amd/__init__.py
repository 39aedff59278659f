"""vfind_amd -- MI355X-native drop-in for vFind's variant-extraction hot path.

Host-side mirror of the reference's only public interface, the PyO3 function `find_variants`
(reference: src/lib.rs:168-232), over the C ABI of include/vfind_hip.h.  The per-read work --
adapter match, semi-global alignment fallback, extraction, translation, counting -- runs in
hand-written HIP kernels (vfind_amd/csrc/vf_k_*.hip); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import VfindError, VfParams, VfStats, check

__all__ = ["find_variants", "Engine", "LocalGroup", "VfindError", "build"]
__version__ = "0.1.0"

build = _lib.build


def _as_bytes(x) -> bytes:
    if isinstance(x, str):
        return x.encode("utf-8")
    if isinstance(x, (bytes, bytearray, memoryview)):
        return bytes(x)
    raise TypeError("adapter must be str")


class Engine:
    """One vf_ctx: a GPU, its count table and its batch pipeline.

    Keyword meaning, defaults and threshold validation follow find_variants (src/lib.rs:169-182,
    100-110, 260-261).  Extra keywords place the context on a device and size its table.
    """

    def __init__(self, adapters: Tuple[str, str], match_score: int = 3, mismatch_score: int = -2,
                 gap_open_penalty: int = 5, gap_extend_penalty: int = 2,
                 accept_prefix_alignment: float = 0.75, accept_suffix_alignment: float = 0.75,
                 n_threads: int = 3, queue_len: int = 2, skip_translation: bool = False, *,
                 device: int = 0, table_slots: int = 0, arena_bytes: int = 0, batch_reads: int = 0):
        if not isinstance(adapters, (tuple, list)) or len(adapters) != 2:
            raise TypeError("adapters must be a tuple (prefix, suffix)")
        self._L = _lib.lib()
        self._pre, self._suf = _as_bytes(adapters[0]), _as_bytes(adapters[1])
        for name, v in (("n_threads", n_threads), ("queue_len", queue_len)):
            if int(v) < 0:
                raise OverflowError(f"can't convert negative int to unsigned ({name})")
        p = VfParams(self._pre, len(self._pre), self._suf, len(self._suf), int(match_score), int(mismatch_score),
                     int(gap_open_penalty), int(gap_extend_penalty), float(accept_prefix_alignment),
                     float(accept_suffix_alignment), int(n_threads), int(queue_len), int(bool(skip_translation)),
                     int(device), int(table_slots), int(arena_bytes), int(batch_reads), 0)
        self._ctx = C.c_void_p()
        check(self._L.vf_ctx_create(C.byref(p), C.byref(self._ctx)))
        self.device = int(device)
        self.skip_translation = bool(skip_translation)

    # ---- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._L.vf_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- submission -----------------------------------------------------------------------------
    def run_fastq(self, path: str):
        """gzip FASTQ file -> count table (replaces lib.rs:233-234 + parallel_fastq, :271-308)"""
        check(self._L.vf_run_fastq(self._ctx, os.fsencode(path)))

    @staticmethod
    def _pack_list(reads) -> Tuple[np.ndarray, np.ndarray]:
        reads = [_as_bytes(r) for r in reads]
        offs = np.zeros(len(reads) + 1, dtype=np.uint64)
        if reads:
            offs[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
        data = np.frombuffer(b"".join(reads), dtype=np.uint8) if reads else np.zeros(0, np.uint8)
        return np.ascontiguousarray(data), offs

    def submit(self, reads_or_arrays):
        """a record set held in host memory: list of sequences, or (uint8 bytes, uint64 offsets)"""
        data, offs = reads_or_arrays if isinstance(reads_or_arrays, tuple) else self._pack_list(reads_or_arrays)
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        check(self._L.vf_submit_ascii(self._ctx, data.ctypes.data, offs.ctypes.data, len(offs) - 1))

    def submit_packed(self, rows: np.ndarray, lens: np.ndarray):
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        lens = np.ascontiguousarray(lens, dtype=np.uint16)
        check(self._L.vf_submit_packed(self._ctx, rows.ctypes.data, rows.shape[1], lens.ctypes.data, rows.shape[0]))

    def batch_acquire(self, n_reads: int, row_dwords: int):
        """page-locked staging of a free batch slot: (handle, rows uint32 [n, W], lens uint16 [n]) as numpy views of the
        library's memory; fill them, then batch_submit(handle, n) -- the views are invalid afterwards"""
        h, rows, lens = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(self._L.vf_batch_acquire(self._ctx, n_reads, row_dwords, C.byref(h), C.byref(rows), C.byref(lens)))
        r = np.ctypeslib.as_array(C.cast(rows, C.POINTER(C.c_uint32)), (n_reads, row_dwords))
        l = np.ctypeslib.as_array(C.cast(lens, C.POINTER(C.c_uint16)), (n_reads,))
        return h, r, l

    def batch_submit(self, handle, n_reads: int):
        check(self._L.vf_batch_submit(self._ctx, handle, n_reads))

    def batch_release(self, handle):
        check(self._L.vf_batch_release(self._ctx, handle))

    def process_device(self, rows_ptr: int, row_dwords: int, lens_ptr: int, n_reads: int, spans_ptr: int = 0):
        """packed batch resident in HBM (raw device addresses, e.g. tensor.data_ptr())"""
        check(self._L.vf_process_device_packed(self._ctx, rows_ptr, row_dwords, lens_ptr, n_reads, spans_ptr or None))

    def spans(self, reads_or_arrays) -> Tuple[np.ndarray, np.ndarray]:
        """per-read find_adapter_match results (lib.rs:278-286) from the device path; -1 = None"""
        data, offs = reads_or_arrays if isinstance(reads_or_arrays, tuple) else self._pack_list(reads_or_arrays)
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        st = np.empty(n, np.int32)
        en = np.empty(n, np.int32)
        check(self._L.vf_spans_ascii(self._ctx, data.ctypes.data, offs.ctypes.data, n, st.ctypes.data, en.ctypes.data))
        return st, en

    def sync(self):
        check(self._L.vf_sync(self._ctx))

    def clear(self):
        check(self._L.vf_table_clear(self._ctx))

    # ---- results --------------------------------------------------------------------------------
    def finish_arrays(self):
        """(key bytes uint8, offsets uint64[rows+1], counts uint64[rows]); count desc, sequence asc"""
        res = C.c_void_p()
        check(self._L.vf_finish(self._ctx, C.byref(res)))
        try:
            n = self._L.vf_result_rows(res)
            offs = np.ctypeslib.as_array(C.cast(self._L.vf_result_key_offsets(res), C.POINTER(C.c_uint64)), (n + 1,)).copy()
            cnts = (np.ctypeslib.as_array(C.cast(self._L.vf_result_counts(res), C.POINTER(C.c_uint64)), (n,)).copy()
                    if n else np.zeros(0, np.uint64))
            nb = int(offs[-1])
            keys = (np.ctypeslib.as_array(C.cast(self._L.vf_result_key_bytes(res), C.POINTER(C.c_uint8)), (nb,)).copy()
                    if nb else np.zeros(0, np.uint8))
        finally:
            self._L.vf_result_free(res)
        return keys, offs, cnts

    def finish_blobs_dict(self, blobs) -> dict:
        """{key: count} of several exported tables [(device pointer, nbytes), ...] summed by the device result build"""
        n = len(blobs)
        ptrs = (C.c_void_p * n)(*[b[0] for b in blobs])
        sizes = (C.c_uint64 * n)(*[b[1] for b in blobs])
        res = C.c_void_p()
        check(self._L.vf_finish_blobs(self._ctx, ptrs, sizes, n, C.byref(res)))
        try:
            rows = self._L.vf_result_rows(res)
            offs = np.ctypeslib.as_array(C.cast(self._L.vf_result_key_offsets(res), C.POINTER(C.c_uint64)), (rows + 1,)).copy()
            cnts = np.ctypeslib.as_array(C.cast(self._L.vf_result_counts(res), C.POINTER(C.c_uint64)), (rows,)).copy() if rows else np.zeros(0, np.uint64)
            nb = int(offs[-1])
            raw = np.ctypeslib.as_array(C.cast(self._L.vf_result_key_bytes(res), C.POINTER(C.c_uint8)), (nb,)).tobytes() if nb else b""
        finally:
            self._L.vf_result_free(res)
        out = {}
        order = []
        for i in range(rows):
            k = raw[int(offs[i]):int(offs[i + 1])]
            assert k not in out, "the device result build left a duplicate row"
            out[k] = int(cnts[i])
            order.append((k, int(cnts[i])))
        assert order == sorted(order, key=lambda kv: (-kv[1], kv[0])), "rows are not ordered by count desc, sequence asc"
        return out

    def finish_dict(self) -> dict:
        keys, offs, cnts = self.finish_arrays()
        raw = keys.tobytes()
        return {raw[int(offs[i]):int(offs[i + 1])]: int(cnts[i]) for i in range(len(cnts))}

    def finish(self):
        """the (sequence, count) frame of lib.rs:312-319: polars.DataFrame if polars is importable, else
        a pyarrow.Table; the buffers come straight from the library through the Arrow C Data Interface"""
        try:
            import pyarrow as pa
        except ImportError:                    # polars alone is enough (the reference needs nothing else): plain arrays
            return _to_frame(*self.finish_arrays())

        res = C.c_void_p()
        check(self._L.vf_finish(self._ctx, C.byref(res)))
        try:
            arr = (C.c_uint64 * 10)()          # struct ArrowArray
            sch = (C.c_uint64 * 9)()           # struct ArrowSchema
            check(self._L.vf_result_export_arrow(res, C.addressof(arr), C.addressof(sch)))
            try:
                batch = pa.RecordBatch._import_from_c(C.addressof(arr), C.addressof(sch))
            except BaseException:
                # the exported structures own the buffers now: release them as a consumer would (C Data Interface:
                # `release` is member 8 of ArrowArray and member 7 of ArrowSchema, null once released)
                rel = C.CFUNCTYPE(None, C.c_void_p)
                if arr[8]:
                    rel(arr[8])(C.addressof(arr))
                if sch[7]:
                    rel(sch[7])(C.addressof(sch))
                raise
        finally:
            self._L.vf_result_free(res)
        try:
            import polars as pl
        except ImportError:
            return pa.Table.from_batches([batch])
        return pl.from_arrow(batch)

    # ---- multi-GPU exchange ----------------------------------------------------------------------
    def export_size(self) -> int:
        n = C.c_uint64()
        check(self._L.vf_table_export_size(self._ctx, C.byref(n)))
        return int(n.value)

    def export_to(self, dst_ptr: int, capacity: int) -> int:
        w = C.c_uint64()
        check(self._L.vf_table_export(self._ctx, dst_ptr, capacity, C.byref(w)))
        return int(w.value)

    def merge_from(self, src_ptr: int, nbytes: int):
        check(self._L.vf_table_merge(self._ctx, src_ptr, nbytes))

    def wait_stream(self, hip_stream: int = 0):
        """everything queued so far on the caller's stream (a raw hipStream_t, e.g. torch.cuda.current_stream().cuda_stream;
        0 = the null stream) becomes a dependency of what this context does next with caller memory (vf_wait_stream)"""
        check(self._L.vf_wait_stream(self._ctx, C.c_void_p(hip_stream or None)))

    def partition_sizes(self, world: int):
        """byte size of every owner's share of this table in an exchange between `world` GPUs"""
        out = (C.c_uint64 * world)()
        check(self._L.vf_table_partition_sizes(self._ctx, world, out))
        return [int(x) for x in out]

    def partition_export(self, world: int, owner: int, dst_ptr: int, capacity: int) -> int:
        w = C.c_uint64()
        check(self._L.vf_table_partition_export(self._ctx, world, owner, dst_ptr, capacity, C.byref(w)))
        return int(w.value)

    @staticmethod
    def comm_unique_id() -> bytes:
        """128-byte RCCL unique id (rank 0 draws it, the host shares it with the other ranks)"""
        buf = C.create_string_buffer(128)
        check(_lib.lib().vf_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        if len(unique_id) != 128:
            raise ValueError("unique_id must be 128 bytes")
        check(self._L.vf_comm_init(self._ctx, int(rank), int(world), unique_id))

    def allreduce_rccl(self, root: int = 0):
        """owner-partitioned reduce of all ranks' tables inside the library (vf_table_allreduce): over RCCL after comm_init,
        between the contexts of a LocalGroup after comm_init_local"""
        check(self._L.vf_table_allreduce(self._ctx, int(root)))

    def comm_init_local(self, rank: int, group: "LocalGroup"):
        """joins an in-process group (vf_comm_init_local): the exchange then runs between contexts of this process"""
        check(self._L.vf_comm_init_local(self._ctx, int(rank), group.handle))
        self._local_group = group                      # keeps the rendezvous alive as long as the context may use it

    def allreduce_table(self, group=None, dst: int = 0, transport: str = "auto"):
        """Sum the (key,count) tables of all ranks of a torch.distributed group at rank `dst`.

        transport "rccl": the library's own exchange over RCCL (vf_comm_init / vf_table_allreduce through the C ABI; the
        128-byte unique id travels through one torch broadcast).  "torch": the same owner-partitioned algorithm with
        torch.distributed moving the bytes (what a CPU-side rehearsal over gloo uses).  "auto": rccl when the group's
        backend is nccl, torch otherwise.  Returns the seconds the exchange took on this rank.
        """
        import time

        import torch
        import torch.distributed as dist

        from .dist import exchange_shares

        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        if world == 1:
            return 0.0
        auto = transport == "auto"
        if auto:
            transport = "rccl" if dist.get_backend(group) == "nccl" else "torch"
        dev = torch.device("cuda", self.device)
        wire = dev if dist.get_backend(group) == "nccl" else torch.device("cpu")
        t0 = time.perf_counter()
        self.last_merge_transport = transport
        if transport == "rccl":
            if not getattr(self, "_comm_ready", False):
                # the communicator first, and every rank hears whether every rank got one: with "auto" a rank that
                # cannot bring RCCL up through the library (no librccl.so.1, an init error) sends everybody to the torch
                # transport before any table has been touched, instead of leaving its peers inside a collective
                err = None
                my_id = None
                try:
                    my_id = self.comm_unique_id()                  # also the probe: loads librccl.so.1 on this rank
                except Exception as e:                            # noqa: BLE001 - agreed on below
                    err = e
                can = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=wire)
                dist.all_reduce(can, op=dist.ReduceOp.MIN, group=group)
                if int(can.item()) == 1:                          # every rank has the library: the collective init is safe to enter
                    uid = torch.zeros(128, dtype=torch.uint8, device=wire)
                    if rank == 0:
                        uid.copy_(torch.frombuffer(bytearray(my_id), dtype=torch.uint8))
                    dist.broadcast(uid, src=0, group=group)
                    try:
                        self.comm_init(rank, world, bytes(uid.cpu().numpy().tobytes()))
                    except Exception as e:                        # noqa: BLE001
                        err = e
                elif err is None:
                    err = VfindError("a peer cannot load RCCL through the library")
                ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=wire)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
                if int(ok.item()) == 0:
                    if not auto:
                        raise err if err is not None else VfindError("a peer could not initialise RCCL through the library")
                    transport = "torch"
                    self.last_merge_transport = f"torch (rccl unavailable{': ' + str(err) if err is not None else ' on a peer'})"
                else:
                    self._comm_ready = True
        if transport == "rccl":
            self.allreduce_rccl(dst)
        else:
            exchange_shares(self, group, dst)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    # ---- stats ----------------------------------------------------------------------------------
    def set_profiling(self, on):
        """0/False off, 1/True events around every kernel, 2 around the table kernel only"""
        check(self._L.vf_set_profiling(self._ctx, int(on)))

    def stats(self) -> dict:
        s = VfStats()
        check(self._L.vf_stats_get(self._ctx, C.byref(s)))
        return {n: getattr(s, n) for n, _ in VfStats._fields_}

    def reset_stats(self):
        check(self._L.vf_stats_reset(self._ctx))


class LocalGroup:
    """rendezvous of `world` contexts of one process (vf_comm_local_create): every context joins with comm_init_local and
    calls allreduce_rccl from a thread of its own (`reduce` does that for a list of engines)"""

    def __init__(self, world: int):
        self._L = _lib.lib()
        h = C.c_void_p()
        check(self._L.vf_comm_local_create(int(world), C.byref(h)))
        self.handle, self.world = h, int(world)

    def close(self):
        if self.handle:
            self._L.vf_comm_local_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    def reduce(self, engines, root: int = 0):
        """vf_table_allreduce on every engine at once, one thread each; returns the per-rank exceptions (None = fine)"""
        import threading
        assert len(engines) == self.world
        errs = [None] * self.world

        def run(r):
            try:
                engines[r].comm_init_local(r, self)
                engines[r].allreduce_rccl(root)
            except Exception as e:                      # noqa: BLE001 - handed to the caller, rank by rank
                errs[r] = e
        th = [threading.Thread(target=run, args=(r,)) for r in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return errs


def _to_frame(keys: np.ndarray, offs: np.ndarray, cnts: np.ndarray):
    raw = keys.tobytes()
    seqs = [raw[int(offs[i]):int(offs[i + 1])].decode("utf-8") for i in range(len(cnts))]
    try:
        import polars as pl
    except ImportError:
        import pyarrow as pa
        return pa.table({"sequence": pa.array(seqs, type=pa.string()), "count": pa.array(cnts, type=pa.uint64())})
    return pl.DataFrame({"sequence": pl.Series(seqs, dtype=pl.String), "count": pl.Series(cnts, dtype=pl.UInt64)})


def find_variants(fq_path: str, adapters: Tuple[str, str], match_score: int = 3, mismatch_score: int = -2,
                  gap_open_penalty: int = 5, gap_extend_penalty: int = 2, accept_prefix_alignment: float = 0.75,
                  accept_suffix_alignment: float = 0.75, n_threads: int = 3, queue_len: int = 2,
                  skip_translation: bool = False, show_progress: bool = True, *, device: int = 0,
                  table_slots: int = 0, arena_bytes: int = 0):
    """Find variable regions flanked by adapters in a FASTQ dataset.

    Same positional/keyword signature, defaults and errors as the reference's `find_variants`
    (src/lib.rs:169-182, 219-232): a bad threshold raises ValueError("Accept alignment threshold
    must be between 0 and 1."), a missing file raises FileNotFoundError.  `show_progress` is
    accepted and ignored (the reference's spinner is never ticked, lib.rs:265-269, 310).

    Returns a frame with columns `sequence` (String) and `count` (UInt64): polars.DataFrame when
    polars is importable, else a pyarrow.Table of the same schema.  Row order (unspecified in the
    reference, lib.rs:312) is count descending, then sequence ascending.
    """
    del show_progress
    with open(fq_path, "rb"):          # File::open(fq_path)? comes first in the reference (lib.rs:233)
        pass
    with Engine(adapters, match_score, mismatch_score, gap_open_penalty, gap_extend_penalty,
                accept_prefix_alignment, accept_suffix_alignment, n_threads, queue_len, skip_translation,
                device=device, table_slots=table_slots, arena_bytes=arena_bytes) as eng:
        eng.run_fastq(fq_path)
        return eng.finish()
