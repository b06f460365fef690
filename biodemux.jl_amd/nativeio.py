"""Native host-side FASTQ reader / packer / in-order demux writer (csrc/bdx_io.cpp, libbdx_io.so).

SURVEY.md §8(f) rank 1 — the callers either side of the hot path (reader_task / writer_task,
core.jl:43-110, :118-224).  Same observable behaviour as the pure-Python path in core.py (which
stays as the reference implementation the tests cross-check against); the hot path in between is
the same single C-ABI call per batch.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

from .classification import filename_for

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "bdx_io.cpp")
LIB_PATH = os.path.join(_HERE, "csrc", "libbdx_io.so")
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(_SRC):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", LIB_PATH, _SRC, "-lz"])
    return LIB_PATH


def available() -> bool:
    return os.path.exists(LIB_PATH)


def _load():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.bdx_io_last_error.restype = C.c_char_p
        L.bdx_fq_open.restype = C.c_int32
        L.bdx_fq_open.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.bdx_fq_open_mt.restype = C.c_int32
        L.bdx_fq_open_mt.argtypes = [C.c_char_p, C.c_int32, C.POINTER(vp)]
        L.bdx_fq_parallel_inflate.restype = C.c_int32
        L.bdx_fq_parallel_inflate.argtypes = [vp]
        L.bdx_fq_close.restype = None
        L.bdx_fq_close.argtypes = [vp]
        L.bdx_fq_size.restype = C.c_int64
        L.bdx_fq_size.argtypes = [vp]
        L.bdx_fq_index.restype = C.c_int64
        L.bdx_fq_index.argtypes = [vp, C.c_int64, C.c_int64, vp, vp, C.POINTER(C.c_int64), C.c_int32]
        L.bdx_fq_release.restype = None
        L.bdx_fq_release.argtypes = [vp, C.c_int64]
        L.bdx_fq_seq_bytes.restype = C.c_int64
        L.bdx_fq_seq_bytes.argtypes = [vp, C.c_int64]
        L.bdx_fq_pack.restype = None
        L.bdx_fq_pack.argtypes = [vp, vp, vp, C.c_int64, vp, vp, C.c_int32]
        L.bdx_fq_demux_write.restype = C.c_int32
        L.bdx_fq_demux_write.argtypes = [vp, vp, vp, C.c_int64, vp, C.c_int32, C.POINTER(C.c_char_p), vp, vp,
                                         C.c_int32, C.c_int32, C.c_int32]
        L.bdx_fq_demux_write_range.restype = C.c_int32
        L.bdx_fq_demux_write_range.argtypes = [vp, vp, vp, C.c_int64, vp, C.c_int32, C.POINTER(C.c_char_p), vp, vp,
                                               C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        _lib = L
    return _lib


def _threads() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, 32))


class FastqFile:
    def __init__(self, path: str, nthreads: int = 0):
        """``nthreads``: inflate threads for size-tagged .gz member chains (BGZF, this library's own output);
        0 = one per available core (at most 16).  Ordinary gzip streams are inflated by one thread."""
        self.L = _load()
        h = C.c_void_p()
        if self.L.bdx_fq_open_mt(path.encode(), int(nthreads), C.byref(h)) != 0:
            raise OSError(self.L.bdx_io_last_error().decode())
        self.h = h
        self.cursor = 0

    @property
    def size(self) -> int:
        """Total bytes (a .gz input is inflated in the background: this waits for the end of the stream)."""
        return int(self.L.bdx_fq_size(self.h))

    @property
    def parallel_inflate(self) -> bool:
        """True when the .gz input was a size-tagged member chain inflated block-parallel (waits for the end)."""
        return bool(self.L.bdx_fq_parallel_inflate(self.h))

    def release(self, upto: int) -> None:
        """Records below byte `upto` are written out: a streamed .gz input gives their pages back."""
        self.L.bdx_fq_release(self.h, int(upto))

    def next_batch(self, max_reads: int, nthreads: int, off=None, ln=None):
        """-> (n_records, line_off int64[4n], line_len int32[4n]) from the cursor on.  ``off`` / ``ln``: arrays to reuse
        (a fresh 48 MB pair per 2^20-read batch is page-faulted in by the indexer)."""
        if off is None or len(off) < 4 * max_reads:
            off = np.empty(4 * max_reads, dtype=np.int64)
        if ln is None or len(ln) < 4 * max_reads:
            ln = np.empty(4 * max_reads, dtype=np.int32)
        nxt = C.c_int64(0)
        n = int(self.L.bdx_fq_index(self.h, self.cursor, max_reads, off.ctypes.data, ln.ctypes.data, C.byref(nxt),
                                    nthreads))
        if n < 0:
            raise OSError(self.L.bdx_io_last_error().decode())
        self.cursor = int(nxt.value)
        return n, off, ln

    def pack(self, off, ln, n: int, nthreads: int, seq=None, so=None):
        total = int(self.L.bdx_fq_seq_bytes(ln.ctypes.data, n))
        if seq is None or len(seq) < max(total, 1):
            seq = np.empty(max(total, 1) + (max(total, 1) >> 4), dtype=np.uint8)
        if so is None or len(so) < n + 1:
            so = np.empty(n + 1, dtype=np.int64)
        self.L.bdx_fq_pack(self.h, off.ctypes.data, ln.ctypes.data, n, seq.ctypes.data, so.ctypes.data, nthreads)
        return (seq[:total] if total else np.zeros(0, dtype=np.uint8)), so[:n + 1]

    def close(self):
        if self.h:
            self.L.bdx_fq_close(self.h)
            self.h = None


# Batch buffers of finished runs (line tables, the packed chunk, verdict vectors: ~120 MB per set at 2^19 reads of 150 bases,
# five sets in flight).  A run takes its sets from here and puts them back: the next execute_demultiplexing of the process
# neither allocates nor page-faults them again, and a run's return does not spend 30 ms giving 600 MB back to the kernel
# (measured: 10 M reads, 0.326 s per call of which 0.030 s after the last batch was written).  `release_buffers()` empties it.
_BUFFER_POOL: list = []
_BUFFER_POOL_MAX = 5


def release_buffers() -> None:
    """Drop the batch buffers kept from finished runs (a long-running host that is done demultiplexing)."""
    del _BUFFER_POOL[:]


def demux_native(fastq1: str, fastq2: Optional[str], config, output_directory: str, prefix1: str, prefix2: str,
                 classifier, batch_reads: int, on_batch=None, timings: Optional[dict] = None) -> None:
    """Native counterpart of core._demux: index -> pack -> ONE C-ABI classify call -> in-order write,
    as a three-stage pipeline (reader thread | classify on the calling thread | writer thread; the
    native calls release the GIL).  Batches flow through bounded FIFO queues, so per-file order is
    input order exactly as with the reference's single writer task (core.jl:139-148)."""
    import queue
    import threading
    import time

    busy = {"index_s": 0.0, "pack_s": 0.0, "classify_s": 0.0, "write_s": 0.0, "batches": 0}
    t_wall = time.perf_counter()
    L = _load()
    T = _threads()
    # (developer knobs: host threads of the reader's / the writer's parallel sections — both default to the whole quota)
    TR = max(1, int(os.environ.get("BDX_IO_READER_THREADS", T)))
    TW = max(1, int(os.environ.get("BDX_IO_WRITER_THREADS", T)))
    f1 = FastqFile(fastq1)
    f2 = FastqFile(fastq2) if fastq2 is not None else None
    stride = max(1, len(config.bc_seqs2)) if config.is_dual else 1
    n_classes = 2 + len(config.bc_seqs) * stride
    do_trim = config.trim_side is not None or config.trim_side2 is not None
    gz = int(bool(config.gzip_output))
    q_in: "queue.Queue" = queue.Queue(maxsize=1)
    errors = []
    # Batch buffers travel reader -> classify -> writer and come back through `free`: line tables, the packed chunk and
    # the verdict vectors are allocated a handful of times per run, not once per batch (fresh arrays of this size are
    # page-faulted in by whoever writes them first: ~15 % of the reader's time)
    free: "queue.Queue" = queue.Queue()
    bufs = [(_BUFFER_POOL.pop() if _BUFFER_POOL else {}) for _ in range(5)]  # one per stage (reader, classify, writer) + one waiting in front of each of the two consumers
    for b in bufs:
        free.put(b)

    def reader():
        try:
            while True:
                # (a batch dropped on an error path never comes back through `free`: poll, and stop once anything failed —
                # the reader must always reach its final q_in.put(None), or the caller waits for it forever)
                while True:
                    try:
                        buf = free.get(timeout=0.05)
                        break
                    except queue.Empty:
                        if errors:
                            return
                if errors:
                    return
                t0 = time.perf_counter()
                n1, off1, ln1 = f1.next_batch(batch_reads, TR, buf.get("off1"), buf.get("ln1"))
                buf["off1"], buf["ln1"] = off1, ln1
                off2 = ln2 = None
                if f2 is not None:
                    n2, off2, ln2 = f2.next_batch(batch_reads, TR, buf.get("off2"), buf.get("ln2"))
                    buf["off2"], buf["ln2"] = off2, ln2
                    n = min(n1, n2)  # lock-step pairs: stop at the shorter file (core.jl:48)
                    last = n1 != n2
                else:
                    n, last = n1, False
                if n == 0:
                    break
                t1 = time.perf_counter()
                seq, so = f1.pack(off1, ln1, n, TR, buf.get("seq"), buf.get("so"))
                if seq.base is not None:
                    buf["seq"] = seq.base
                if so.base is not None:
                    buf["so"] = so.base
                busy["index_s"] += t1 - t0
                busy["pack_s"] += time.perf_counter() - t1
                q_in.put((n, off1, ln1, off2, ln2, seq, so, f1.cursor, f2.cursor if f2 is not None else 0, buf))
                if last:
                    break
        except BaseException as e:  # noqa: BLE001 - forwarded to the caller
            errors.append(e)
        finally:
            q_in.put(None)

    def paths(prefix, used):
        arr = (C.c_char_p * n_classes)()
        for c in used:
            c = int(c)
            if c == 0:
                b1, b2 = 0, 0
            elif c == 1:
                b1, b2 = -1, 0
            else:
                b1, b2 = divmod(c - 2, stride)
                b1, b2 = b1 + 1, (b2 + 1 if config.is_dual else 0)
            arr[c] = os.path.join(output_directory, prefix + "." + filename_for(config, b1, b2)).encode()
        return arr

    # The writer gathers every batch straight into the (mapped) output files with all its threads working on all files
    # (csrc/bdx_io.cpp): one writer thread, one class range.  (bdx_fq_demux_write_range lets several writer threads
    # share a batch by class range — each file then still belongs to exactly one of them, in batch order: per-file order
    # = input order, core.jl:139-148 — which pays on file systems without shared writable mappings.)
    ranges = [(0, n_classes)]
    q_outs = [queue.Queue(maxsize=1) for _ in ranges]
    done_lock = threading.Lock()

    def writer(wi):
        lo, hi = ranges[wi]
        qw = q_outs[wi]
        tshare = TW
        try:
            while True:
                item = qw.get()
                if item is None:
                    break
                n, off1, ln1, off2, ln2, cls, ks, ke, cur1, cur2, buf, pending, used = item
                tw0 = time.perf_counter()

                def write(f, off, ln, prefix, trim):
                    rc = L.bdx_fq_demux_write_range(f.h, off.ctypes.data, ln.ctypes.data, n, cls.ctypes.data, n_classes,
                                                    paths(prefix, used), ks.ctypes.data, ke.ctypes.data, int(trim), gz, tshare, lo, hi)
                    if rc != 0:
                        raise OSError(L.bdx_io_last_error().decode())

                if used[(used >= lo) & (used < hi)].size:
                    if config.classify_both and f2 is not None:  # core.jl:175-185
                        write(f1, off1, ln1, prefix1, do_trim)
                        write(f2, off2, ln2, prefix2, False)
                    elif f2 is not None:  # core.jl:186-190
                        write(f2, off2, ln2, prefix2, False)
                    else:  # core.jl:191-196
                        write(f1, off1, ln1, prefix1, do_trim)
                with done_lock:
                    pending[0] -= 1
                    last = pending[0] == 0
                    busy["write_s"] += time.perf_counter() - tw0
                if last:
                    f1.release(cur1)  # this batch and everything before it is on disk
                    if f2 is not None:
                        f2.release(cur2)
                    free.put(buf)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            while True:  # keep draining so the producer never blocks; the dropped batches' buffers go back
                item = qw.get()
                if item is None:
                    break
                free.put(item[10])

    tr = threading.Thread(target=reader, name="bdx-reader")
    tws = [threading.Thread(target=writer, args=(wi,), name=f"bdx-writer-{wi}") for wi in range(len(ranges))]
    tr.start()
    for tw in tws:
        tw.start()
    try:
        while True:
            item = q_in.get()
            if item is None:
                break
            if errors:
                free.put(item[-1])
                continue
            n, off1, ln1, off2, ln2, seq, so, cur1, cur2, buf = item
            tc0 = time.perf_counter()
            res = buf.get("out")
            reuse = (res is not None and len(res["bc1"]) >= n and getattr(classifier, "want_pass", False) is False
                     and hasattr(classifier, "lib"))  # (the HIP wrapper takes result arrays to reuse; test doubles may not)
            if reuse:
                out = classifier.classify(seq, so, out={k: v[:n] for k, v in res.items()})  # <- the hot path: one C-ABI call per batch
            else:
                out = classifier.classify(seq, so)
                if set(out) == {"bc1", "bc2", "keep_start", "keep_end"}:
                    buf["out"] = out
            busy["classify_s"] += time.perf_counter() - tc0
            busy["batches"] += 1
            if on_batch is not None:
                on_batch(out)
            bc1, bc2 = out["bc1"], out["bc2"]
            cls = buf.get("cls")
            if cls is None or len(cls) < n:
                cls = buf["cls"] = np.empty(max(n, batch_reads), dtype=np.int32)
            cls = cls[:n]
            # class of a read: 0 unknown, 1 ambiguous, 2 + (bc1 - 1) * stride + (bc2 - 1) matched
            np.subtract(bc1, 1, out=cls)
            if stride != 1:
                np.multiply(cls, stride, out=cls)
                cls += np.maximum(bc2, 1)
                cls += 1
            else:
                cls += 2
            cls[bc1 == 0] = 0
            cls[bc1 < 0] = 1
            ks = np.ascontiguousarray(out["keep_start"], dtype=np.int32)
            ke = np.ascontiguousarray(out["keep_end"], dtype=np.int32)
            used = np.flatnonzero(np.bincount(cls, minlength=n_classes))
            pending = [len(ranges)]
            for qw in q_outs:
                qw.put((n, off1, ln1, off2, ln2, cls, ks, ke, cur1, cur2, buf, pending, used))
    except BaseException as e:  # noqa: BLE001
        errors.append(e)
        while True:
            item = q_in.get()
            if item is None:
                break
            free.put(item[-1])
    finally:
        for qw in q_outs:
            qw.put(None)
        tr.join()
        for tw in tws:
            tw.join()
        f1.close()
        if f2 is not None:
            f2.close()
        for b in bufs:  # (every thread has been joined: nobody holds a set any more)
            if len(_BUFFER_POOL) < _BUFFER_POOL_MAX:
                _BUFFER_POOL.append(b)
    if timings is not None:  # busy seconds of the three overlapped stages (reader = index + pack | classify | writer)
        busy["wall_s"] = time.perf_counter() - t_wall
        busy["threads"] = T
        timings.update(busy)
    if errors:
        raise errors[0]
