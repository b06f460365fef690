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

    def next_batch(self, max_reads: int, nthreads: int):
        """-> (n_records, line_off int64[4n], line_len int32[4n]) from the cursor on."""
        off = np.empty(4 * max_reads, dtype=np.int64)
        ln = np.empty(4 * max_reads, dtype=np.int32)
        nxt = C.c_int64(0)
        n = int(self.L.bdx_fq_index(self.h, self.cursor, max_reads, off.ctypes.data, ln.ctypes.data, C.byref(nxt),
                                    nthreads))
        if n < 0:
            raise OSError(self.L.bdx_io_last_error().decode())
        self.cursor = int(nxt.value)
        return n, off, ln

    def pack(self, off, ln, n: int, nthreads: int):
        total = int(self.L.bdx_fq_seq_bytes(ln.ctypes.data, n))
        seq = np.empty(max(total, 1), dtype=np.uint8)
        so = np.empty(n + 1, dtype=np.int64)
        self.L.bdx_fq_pack(self.h, off.ctypes.data, ln.ctypes.data, n, seq.ctypes.data, so.ctypes.data, nthreads)
        return seq[:total] if total else np.zeros(0, dtype=np.uint8), so

    def close(self):
        if self.h:
            self.L.bdx_fq_close(self.h)
            self.h = None


def demux_native(fastq1: str, fastq2: Optional[str], config, output_directory: str, prefix1: str, prefix2: str,
                 classifier, batch_reads: int, on_batch=None) -> None:
    """Native counterpart of core._demux: index -> pack -> ONE C-ABI classify call -> in-order write,
    as a three-stage pipeline (reader thread | classify on the calling thread | writer thread; the
    native calls release the GIL).  Batches flow through bounded FIFO queues, so per-file order is
    input order exactly as with the reference's single writer task (core.jl:139-148)."""
    import queue
    import threading

    L = _load()
    T = _threads()
    f1 = FastqFile(fastq1)
    f2 = FastqFile(fastq2) if fastq2 is not None else None
    stride = max(1, len(config.bc_seqs2)) if config.is_dual else 1
    n_classes = 2 + len(config.bc_seqs) * stride
    do_trim = config.trim_side is not None or config.trim_side2 is not None
    gz = int(bool(config.gzip_output))
    q_in: "queue.Queue" = queue.Queue(maxsize=2)
    q_out: "queue.Queue" = queue.Queue(maxsize=2)
    errors = []

    def reader():
        try:
            while True:
                n1, off1, ln1 = f1.next_batch(batch_reads, T)
                off2 = ln2 = None
                if f2 is not None:
                    n2, off2, ln2 = f2.next_batch(batch_reads, T)
                    n = min(n1, n2)  # lock-step pairs: stop at the shorter file (core.jl:48)
                    last = n1 != n2
                else:
                    n, last = n1, False
                if n == 0:
                    break
                seq, so = f1.pack(off1, ln1, n, T)
                q_in.put((n, off1, ln1, off2, ln2, seq, so, f1.cursor, f2.cursor if f2 is not None else 0))
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

    def writer():
        try:
            while True:
                item = q_out.get()
                if item is None:
                    break
                n, off1, ln1, off2, ln2, cls, ks, ke, cur1, cur2 = item
                used = np.unique(cls)

                def write(f, off, ln, prefix, trim):
                    rc = L.bdx_fq_demux_write(f.h, off.ctypes.data, ln.ctypes.data, n, cls.ctypes.data, n_classes,
                                              paths(prefix, used), ks.ctypes.data, ke.ctypes.data, int(trim), gz, T)
                    if rc != 0:
                        raise OSError(L.bdx_io_last_error().decode())

                if config.classify_both and f2 is not None:  # core.jl:175-185
                    write(f1, off1, ln1, prefix1, do_trim)
                    write(f2, off2, ln2, prefix2, False)
                elif f2 is not None:  # core.jl:186-190
                    write(f2, off2, ln2, prefix2, False)
                else:  # core.jl:191-196
                    write(f1, off1, ln1, prefix1, do_trim)
                f1.release(cur1)  # this batch and everything before it is on disk
                if f2 is not None:
                    f2.release(cur2)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            while q_out.get() is not None:  # keep draining so the producer never blocks
                pass

    tr = threading.Thread(target=reader, name="bdx-reader")
    tw = threading.Thread(target=writer, name="bdx-writer")
    tr.start()
    tw.start()
    try:
        while True:
            item = q_in.get()
            if item is None:
                break
            if errors:
                continue
            n, off1, ln1, off2, ln2, seq, so, cur1, cur2 = item
            out = classifier.classify(seq, so)  # <- the hot path: one C-ABI call per batch
            if on_batch is not None:
                on_batch(out)
            bc1, bc2 = out["bc1"], out["bc2"]
            cls = np.where(bc1 > 0, 2 + (bc1 - 1) * stride + np.maximum(bc2 - 1, 0), np.where(bc1 == 0, 0, 1))
            cls = np.ascontiguousarray(cls, dtype=np.int32)
            ks = np.ascontiguousarray(out["keep_start"], dtype=np.int32)
            ke = np.ascontiguousarray(out["keep_end"], dtype=np.int32)
            q_out.put((n, off1, ln1, off2, ln2, cls, ks, ke, cur1, cur2))
    except BaseException as e:  # noqa: BLE001
        errors.append(e)
        while q_in.get() is not None:
            pass
    finally:
        q_out.put(None)
        tr.join()
        tw.join()
        f1.close()
        if f2 is not None:
            f2.close()
    if errors:
        raise errors[0]
