"""ctypes binding of the C-ABI (include/biodemux_hip.h) and the batch classifier built on it.

This is the Python counterpart of the Julia ``ccall`` shim shown in INTEGRATION.md: it packs a
DemuxConfig (classification.jl:16-58) into ``bdx_config_t`` and calls
``bdx_classify_host`` / ``bdx_classify_device`` once per chunk, replacing the per-read loop of
worker_task (core.jl:243-267).  There is no CPU fallback: if libbiodemux_hip.so is missing or
no HIP device is present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BDX_LIB_PATH") or os.path.join(_HERE, "csrc", "libbiodemux_hip.so")  # (override: test builds)

BDX_ABI_VERSION = 1
ALG = {"semiglobal": 0, "hamming": 1, "exact": 2}
FILTER = {"auto": 0, "off": 1, "qgram": 2, "bitpar": 3}
FILTER_NAMES = {v: k for k, v in FILTER.items()}

# every symbol include/biodemux_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "bdx_abi_version", "bdx_create", "bdx_destroy", "bdx_last_error", "bdx_classify_host",
    "bdx_classify_device", "bdx_sync", "bdx_set_stream", "bdx_counts_len", "bdx_get_counts",
    "bdx_reset_counts", "bdx_counts_device_ptr", "bdx_set_counts_buffer", "bdx_kernel_path",
    "bdx_launch_info", "bdx_set_read_length_hint", "bdx_host_alloc", "bdx_host_free",
    # merge_stats across GPUs (RCCL, opened lazily)
    "bdx_comm_get_unique_id", "bdx_comm_init_rank", "bdx_comm_init_all", "bdx_comm_destroy", "bdx_comm_rank",
    "bdx_comm_size", "bdx_allreduce_counts", "bdx_allreduce_counts_all", "bdx_reduced_counts_device_ptr",
    "bdx_get_reduced_counts",
    # DemuxStats histograms (summary = true), collected on the device
    "bdx_stats_shape", "bdx_get_stats",
    "bdx_window_uploads", "bdx_band_launches", "bdx_wave_launches", "bdx_pair_launches", "bdx_pipelined_calls", "bdx_staged_downloads", "bdx_last_list_reads", "bdx_rejected_windows",
    "bdx_debug_rejected_windows_total",
]
STATS_WHICH = {"pos": 0, "len": 1, "raw": 2}
BDX_COMM_ID_BYTES = 128


def pinned_empty(n: int, dtype) -> "np.ndarray":
    """A numpy array in page-locked host memory (bdx_host_alloc): buffers handed to ``HipClassifier.classify``
    from such arrays are copied by asynchronous DMA.  The memory is released when the array is collected."""
    import weakref

    lib = load_library()
    dt = np.dtype(dtype)
    nbytes = max(1, int(n) * dt.itemsize)
    p = lib.bdx_host_alloc(nbytes)
    if not p:
        raise MemoryError(f"bdx_host_alloc({nbytes}) failed")
    buf = (C.c_uint8 * nbytes).from_address(p)
    arr = np.frombuffer(buf, dtype=dt, count=int(n))
    weakref.finalize(buf, lib.bdx_host_free, p)
    return arr


class BdxError(RuntimeError):
    """Raised for any non-zero return of the C-ABI (the Julia shim calls error(msg))."""


class BdxRange(C.Structure):
    _fields_ = [
        ("start_offset", C.c_int64),
        ("end_offset", C.c_int64),
        ("start_from_end", C.c_int32),
        ("end_from_end", C.c_int32),
    ]


class BdxPass(C.Structure):
    _fields_ = [
        ("ref_search_range", BdxRange),
        ("barcode_start_range", BdxRange),
        ("barcode_end_range", BdxRange),
        ("trim_side", C.c_int32),
        ("n_barcodes", C.c_int32),
        ("bc_bytes", C.POINTER(C.c_uint8)),
        ("bc_off", C.POINTER(C.c_uint32)),
        ("bc_len_no_N", C.POINTER(C.c_int32)),
        ("explicit_window", C.c_int32),
        ("_pad", C.c_int32),
        ("win_first", C.c_int64),
        ("win_last", C.c_int64),
        ("win_max_start_pos", C.c_int64),
        ("win_min_end_pos", C.c_int64),
    ]


class BdxConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("struct_size", C.c_uint32),
        ("algorithm", C.c_int32),
        ("is_dual", C.c_int32),
        ("max_error_rate", C.c_double),
        ("min_delta", C.c_double),
        ("match", C.c_int32),
        ("mismatch", C.c_int32),
        ("indel", C.c_int32),
        ("has_nindel", C.c_int32),
        ("nindel", C.c_int32),
        ("need_traceback", C.c_int32),
        ("filter", C.c_int32),
        ("device", C.c_int32),
        ("pass_", BdxPass * 2),
    ]


class BdxOutputs(C.Structure):
    _fields_ = [
        ("bc1", C.c_void_p),
        ("bc2", C.c_void_p),
        ("keep_start", C.c_void_p),
        ("keep_end", C.c_void_p),
        ("pass_start", C.c_void_p),
        ("pass_end", C.c_void_p),
        ("pass_raw", C.c_void_p),
        ("pass_score", C.c_void_p),
        ("pass_bc", C.c_void_p),
        ("pass_delta", C.c_void_p),
    ]


class BdxLaunchInfo(C.Structure):
    _fields_ = [
        ("threads_per_block", C.c_int32),
        ("lds_bytes_per_block", C.c_int32),
        ("blocks", C.c_int64),
        ("reads_per_block", C.c_int32),
        ("filter_used", C.c_int32),
        ("max_m", C.c_int32),
        ("launches", C.c_int64),
    ]


_lib = None


def load_library(path: Optional[str] = None):
    """dlopen libbiodemux_hip.so and declare the prototypes.  Raises if the library was not
    built (run ``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise BdxError(f"HIP extension not built: {p} is missing (no CPU fallback exists; run __graft_entry__.build())")
    L = C.CDLL(p)
    vp = C.c_void_p
    L.bdx_abi_version.restype = C.c_int32
    L.bdx_abi_version.argtypes = []
    L.bdx_create.restype = C.c_int32
    L.bdx_create.argtypes = [C.POINTER(BdxConfig), C.POINTER(vp)]
    L.bdx_destroy.restype = None
    L.bdx_destroy.argtypes = [vp]
    L.bdx_last_error.restype = C.c_char_p
    L.bdx_last_error.argtypes = [vp]
    L.bdx_classify_host.restype = C.c_int32
    L.bdx_classify_host.argtypes = [vp, vp, vp, C.c_int64, C.POINTER(BdxOutputs)]
    L.bdx_classify_device.restype = C.c_int32
    L.bdx_classify_device.argtypes = [vp, vp, vp, C.c_int64, C.POINTER(BdxOutputs)]
    L.bdx_sync.restype = C.c_int32
    L.bdx_sync.argtypes = [vp]
    L.bdx_set_stream.restype = C.c_int32
    L.bdx_set_stream.argtypes = [vp, vp]
    L.bdx_counts_len.restype = C.c_int64
    L.bdx_counts_len.argtypes = [vp]
    L.bdx_get_counts.restype = C.c_int32
    L.bdx_get_counts.argtypes = [vp, vp, C.c_int64]
    L.bdx_reset_counts.restype = C.c_int32
    L.bdx_reset_counts.argtypes = [vp]
    L.bdx_counts_device_ptr.restype = vp
    L.bdx_counts_device_ptr.argtypes = [vp]
    L.bdx_set_counts_buffer.restype = C.c_int32
    L.bdx_set_counts_buffer.argtypes = [vp, vp]
    L.bdx_set_read_length_hint.restype = C.c_int32
    L.bdx_set_read_length_hint.argtypes = [vp, C.c_int32]
    L.bdx_host_alloc.restype = vp
    L.bdx_host_alloc.argtypes = [C.c_size_t]
    L.bdx_host_free.restype = None
    L.bdx_host_free.argtypes = [vp]
    L.bdx_kernel_path.restype = C.c_char_p
    L.bdx_kernel_path.argtypes = [vp]
    L.bdx_launch_info.restype = C.c_int32
    L.bdx_launch_info.argtypes = [vp, C.POINTER(BdxLaunchInfo)]
    L.bdx_comm_get_unique_id.restype = C.c_int32
    L.bdx_comm_get_unique_id.argtypes = [vp]
    L.bdx_comm_init_rank.restype = C.c_int32
    L.bdx_comm_init_rank.argtypes = [vp, vp, C.c_int32, C.c_int32]
    L.bdx_comm_init_all.restype = C.c_int32
    L.bdx_comm_init_all.argtypes = [C.POINTER(vp), C.c_int32]
    L.bdx_comm_destroy.restype = C.c_int32
    L.bdx_comm_destroy.argtypes = [vp]
    L.bdx_comm_rank.restype = C.c_int32
    L.bdx_comm_rank.argtypes = [vp]
    L.bdx_comm_size.restype = C.c_int32
    L.bdx_comm_size.argtypes = [vp]
    L.bdx_allreduce_counts.restype = C.c_int32
    L.bdx_allreduce_counts.argtypes = [vp]
    L.bdx_allreduce_counts_all.restype = C.c_int32
    L.bdx_allreduce_counts_all.argtypes = [C.POINTER(vp), C.c_int32]
    L.bdx_reduced_counts_device_ptr.restype = vp
    L.bdx_reduced_counts_device_ptr.argtypes = [vp]
    L.bdx_get_reduced_counts.restype = C.c_int32
    L.bdx_get_reduced_counts.argtypes = [vp, vp, C.c_int64]
    L.bdx_stats_shape.restype = C.c_int32
    L.bdx_stats_shape.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.bdx_get_stats.restype = C.c_int32
    L.bdx_get_stats.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int64]
    L.bdx_window_uploads.restype = C.c_int64
    L.bdx_window_uploads.argtypes = [vp]
    L.bdx_band_launches.restype = C.c_int64
    L.bdx_band_launches.argtypes = [vp]
    L.bdx_wave_launches.restype = C.c_int64
    L.bdx_wave_launches.argtypes = [vp]
    L.bdx_pair_launches.restype = C.c_int64
    L.bdx_pair_launches.argtypes = [vp]
    L.bdx_pipelined_calls.restype = C.c_int64
    L.bdx_pipelined_calls.argtypes = [vp]
    L.bdx_staged_downloads.restype = C.c_int64
    L.bdx_staged_downloads.argtypes = [vp]
    L.bdx_last_list_reads.restype = C.c_int64
    L.bdx_last_list_reads.argtypes = [vp]
    L.bdx_rejected_windows.restype = C.c_int64
    L.bdx_rejected_windows.argtypes = [vp]
    L.bdx_debug_rejected_windows_total.restype = C.c_int64
    L.bdx_debug_rejected_windows_total.argtypes = []
    if path is None:
        _lib = L
    return L


def comm_unique_id() -> bytes:
    """bdx_comm_get_unique_id: the 128-byte RCCL id rank 0 makes and the host ships to every rank."""
    lib = load_library()
    buf = (C.c_uint8 * BDX_COMM_ID_BYTES)()
    if lib.bdx_comm_get_unique_id(buf) != 0:
        raise BdxError(lib.bdx_last_error(None).decode())
    return bytes(buf)


def comm_init_all(classifiers) -> None:
    """bdx_comm_init_all: one process, one HipClassifier per (distinct) device."""
    lib = load_library()
    arr = (C.c_void_p * len(classifiers))(*[c.h for c in classifiers])
    if lib.bdx_comm_init_all(arr, len(classifiers)) != 0:
        raise BdxError((lib.bdx_last_error(classifiers[0].h) or lib.bdx_last_error(None)).decode())


def allreduce_counts_all(classifiers) -> None:
    """bdx_allreduce_counts_all: the grouped collective over the contexts of one process."""
    lib = load_library()
    arr = (C.c_void_p * len(classifiers))(*[c.h for c in classifiers])
    if lib.bdx_allreduce_counts_all(arr, len(classifiers)) != 0:
        raise BdxError((lib.bdx_last_error(classifiers[0].h) or lib.bdx_last_error(None)).decode())


def _mk_range(dr) -> BdxRange:
    return BdxRange(int(dr.start_offset), int(dr.end_offset), int(bool(dr.start_from_end)), int(bool(dr.end_from_end)))


def _ts(trim_side) -> int:
    return 0 if trim_side is None else int(trim_side)


def pack_config(cfg, device: int = 0, filter: str = "auto", windows=None):
    """DemuxConfig -> (bdx_config_t, keepalive list).  ``windows`` optionally maps pass index
    to ``(first, last, max_start_pos, min_end_pos[, mode])`` for the unit-level API
    (mode 1 = find_best window, 2 = one direct alignment call; see biodemux_hip.h)."""
    c = BdxConfig()
    c.abi_version = BDX_ABI_VERSION
    c.struct_size = C.sizeof(BdxConfig)
    c.algorithm = ALG[str(cfg.matching_algorithm).lstrip(":")]
    c.is_dual = int(bool(cfg.is_dual))
    c.max_error_rate = float(cfg.max_error_rate)
    c.min_delta = float(cfg.min_delta)
    c.match, c.mismatch, c.indel = int(cfg.match), int(cfg.mismatch), int(cfg.indel)
    c.has_nindel = 0 if cfg.nindel is None else 1
    c.nindel = 0 if cfg.nindel is None else int(cfg.nindel)
    c.need_traceback = int(bool(cfg.summary))
    c.filter = FILTER[filter]
    c.device = int(device)
    keep = []
    passes = [
        (cfg.ref_search_range, cfg.barcode_start_range, cfg.barcode_end_range, cfg.bc_seqs, cfg.bc_lengths_no_N,
         cfg.trim_side),
        (cfg.ref_search_range2, cfg.barcode_start_range2, cfg.barcode_end_range2, cfg.bc_seqs2,
         cfg.bc_lengths_no_N2, cfg.trim_side2),
    ]
    for p, (rs, bs, be, seqs, lens, ts) in enumerate(passes):
        P = c.pass_[p]
        P.ref_search_range = _mk_range(rs)
        P.barcode_start_range = _mk_range(bs)
        P.barcode_end_range = _mk_range(be)
        P.trim_side = _ts(ts)
        P.n_barcodes = len(seqs)
        raw = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs]
        off = np.zeros(len(raw) + 1, dtype=np.uint32)
        if raw:
            off[1:] = np.cumsum([len(b) for b in raw])
        blob = np.frombuffer(b"".join(raw) + b"\0", dtype=np.uint8).copy()
        ln = np.asarray(list(lens) if len(lens) else [0], dtype=np.int32)
        keep += [blob, off, ln]
        P.bc_bytes = blob.ctypes.data_as(C.POINTER(C.c_uint8))
        P.bc_off = off.ctypes.data_as(C.POINTER(C.c_uint32))
        P.bc_len_no_N = ln.ctypes.data_as(C.POINTER(C.c_int32))
        if windows and p in windows:
            w = windows[p]
            P.explicit_window = int(w[4]) if len(w) > 4 else 1
            P.win_first, P.win_last, P.win_max_start_pos, P.win_min_end_pos = (int(x) for x in w[:4])
    return c, keep


class HipClassifier:
    """One C-ABI context: the drop-in for a reference worker (core.jl:226-279).

    ``classify(seq_bytes, seq_off)`` takes the packed chunk (uint8 code units + int64
    offsets) and returns numpy arrays ``bc1, bc2, keep_start, keep_end`` (+ per-pass
    ``pass_start/pass_end/pass_raw/pass_score`` when ``want_pass``), i.e. for every read the
    verdict of determine_filename (classification.jl:871-938).
    """

    def __init__(self, cfg, device: int = 0, filter: str = "auto", want_pass: bool = False, windows=None):
        self.lib = load_library()
        self.cfg = cfg
        c, keep = pack_config(cfg, device=device, filter=filter, windows=windows)
        self._c, self._keep = c, keep
        h = C.c_void_p()
        rc = self.lib.bdx_create(C.byref(c), C.byref(h))
        if rc != 0:
            raise BdxError(self.lib.bdx_last_error(None).decode())
        self.h = h
        self.want_pass = want_pass
        self.device = device

    # -- lifecycle --
    def close(self):
        if getattr(self, "h", None):
            self.lib.bdx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise BdxError(self.lib.bdx_last_error(self.h).decode())

    # -- host-buffer entry point --
    def classify(self, seq_bytes: np.ndarray, seq_off: np.ndarray, out: dict = None) -> dict:
        """One packed chunk through bdx_classify_host.  ``out``: result arrays to (re)use — what a long-running host
        does (the reference's worker keeps its vectors, core.jl:238-241); freshly allocated pageable arrays are
        page-faulted in by the device-to-host copies (13 of 47 ms for four outputs of 10 M reads).  Arrays from
        ``pinned_empty`` make the copies asynchronous DMA."""
        seq_bytes = np.ascontiguousarray(seq_bytes, dtype=np.uint8)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
        n = len(seq_off) - 1
        if out is not None:
            for k, v in out.items():
                want = (n, 2) if k.startswith("pass_") else (n,)
                dt = np.float64 if k in ("pass_score", "pass_delta") else np.int32
                if v.shape != want or v.dtype != dt or not v.flags.c_contiguous:
                    raise ValueError(f"out[{k!r}]: expected a C-contiguous {np.dtype(dt).name} array of shape {want}")
            if "bc1" not in out:
                raise ValueError("out must at least hold 'bc1'")
        else:
            # every entry of every requested output is written by the kernels: no initialisation pass over them
            out = {k: np.empty(n, dtype=np.int32) for k in ("bc1", "bc2", "keep_start", "keep_end")}
            if self.want_pass:
                for k in ("pass_start", "pass_end", "pass_raw", "pass_bc"):
                    out[k] = np.empty((n, 2), dtype=np.int32)
                for k in ("pass_score", "pass_delta"):
                    out[k] = np.empty((n, 2), dtype=np.float64)
        if n == 0:
            return out
        if seq_bytes.size == 0:
            seq_bytes = np.zeros(1, dtype=np.uint8)
        o = BdxOutputs()
        for k in out:
            setattr(o, k, out[k].ctypes.data)
        self._check(self.lib.bdx_classify_host(self.h, seq_bytes.ctypes.data, seq_off.ctypes.data, n, C.byref(o)))
        return out

    # -- device-resident entry point (pointers are raw device addresses, e.g. tensor.data_ptr()) --
    def classify_device(self, d_seq: int, d_off: int, n_reads: int, **d_out: int):
        o = BdxOutputs()
        for k, v in d_out.items():
            setattr(o, k, v)
        self._check(self.lib.bdx_classify_device(self.h, d_seq, d_off, n_reads, C.byref(o)))

    def sync(self):
        self._check(self.lib.bdx_sync(self.h))

    def set_read_length_hint(self, n: int):
        self._check(self.lib.bdx_set_read_length_hint(self.h, int(n)))

    def set_stream(self, hip_stream: int):
        self._check(self.lib.bdx_set_stream(self.h, hip_stream))

    # -- DemuxStats scalar counters --
    @property
    def counts_len(self) -> int:
        return int(self.lib.bdx_counts_len(self.h))

    @property
    def counts(self) -> np.ndarray:
        out = np.zeros(self.counts_len, dtype=np.int64)
        self._check(self.lib.bdx_get_counts(self.h, out.ctypes.data, len(out)))
        return out

    def reset_counts(self):
        self._check(self.lib.bdx_reset_counts(self.h))

    def set_counts_buffer(self, d_ptr: int):
        self._check(self.lib.bdx_set_counts_buffer(self.h, d_ptr))

    # -- merge_stats across GPUs through the C-ABI (RCCL) --
    def comm_init_rank(self, unique_id: bytes, rank: int, n_ranks: int):
        assert len(unique_id) == BDX_COMM_ID_BYTES
        buf = (C.c_uint8 * BDX_COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self.lib.bdx_comm_init_rank(self.h, buf, int(rank), int(n_ranks)))

    def comm_destroy(self):
        self._check(self.lib.bdx_comm_destroy(self.h))

    @property
    def comm_size(self) -> int:
        return int(self.lib.bdx_comm_size(self.h))

    @property
    def comm_rank(self) -> int:
        return int(self.lib.bdx_comm_rank(self.h))

    def allreduce_counts(self):
        """Enqueue the all-reduce (sum over the ranks) of the counter vector on this context's stream."""
        self._check(self.lib.bdx_allreduce_counts(self.h))

    @property
    def reduced_counts(self) -> np.ndarray:
        out = np.zeros(self.counts_len, dtype=np.int64)
        self._check(self.lib.bdx_get_reduced_counts(self.h, out.ctypes.data, len(out)))
        return out

    def stats_tables(self, reduced: bool = False) -> dict:
        """The DemuxStats histograms the device collected (summary = true): ``{pass: {"pos" | "len" | "raw":
        (int64[rows, n_barcodes], key0)}}`` — row r of a table holds key r + key0 (bdx_stats_shape / bdx_get_stats).
        ``reduced`` reads the tables summed over the ranks by allreduce_counts() (rows a later batch appended to the
        per-rank tables since then read as zero)."""
        out = {}
        for p in range(2 if self.cfg.is_dual else 1):
            out[p] = {}
            for name, w in STATS_WHICH.items():
                rows, key0, nb = C.c_int64(), C.c_int64(), C.c_int64()
                self._check(self.lib.bdx_stats_shape(self.h, p, w, C.byref(rows), C.byref(key0), C.byref(nb)))
                tab = np.zeros((max(rows.value, 0), max(nb.value, 0)), dtype=np.int64)
                if tab.size or rows.value:
                    buf = np.zeros(max(tab.size, 1), dtype=np.int64)
                    self._check(self.lib.bdx_get_stats(self.h, p, w, int(bool(reduced)), buf.ctypes.data, len(buf)))
                    tab = buf[:tab.size].reshape(tab.shape)
                out[p][name] = (tab, int(key0.value))
        return out

    @property
    def band_launches(self) -> int:
        """(pass, exact-kernel launch) pairs that ran with the diagonal-band DP enabled."""
        return int(self.lib.bdx_band_launches(self.h))

    @property
    def pipelined_calls(self) -> int:
        """classify() calls that uploaded their batch in chunks beside the previous chunk's kernels."""
        return int(self.lib.bdx_pipelined_calls(self.h))

    @property
    def last_list_reads(self) -> int:
        """Reads the first filter launch of the last classify call handed on through its device-side list."""
        return int(self.lib.bdx_last_list_reads(self.h))

    @property
    def staged_downloads(self) -> int:
        """classify() calls whose result vectors came back through the page-locked staging buffer (large pageable outputs)."""
        return int(self.lib.bdx_staged_downloads(self.h))

    @property
    def rejected_windows(self) -> int:
        """Hand-over windows the exact kernel refused as "not a window" (must be 0; synchronises the stream)."""
        return int(self.lib.bdx_rejected_windows(self.h))

    @property
    def wave_launches(self) -> int:
        """Launches of the wave-autonomous kernel (bdx_wave.hip) this context made."""
        return int(self.lib.bdx_wave_launches(self.h))

    @property
    def pair_launches(self) -> int:
        """Launches of the wave kernel's pairs mode (bdx_pairs.hip) this context made."""
        return int(self.lib.bdx_pair_launches(self.h))

    @property
    def window_uploads(self) -> int:
        """classify() calls that uploaded only each read's column window (long reads, short windows)."""
        return int(self.lib.bdx_window_uploads(self.h))

    @property
    def kernel_path(self) -> str:
        return self.lib.bdx_kernel_path(self.h).decode()

    def launch_info(self) -> dict:
        li = BdxLaunchInfo()
        self._check(self.lib.bdx_launch_info(self.h, C.byref(li)))
        d = {k: getattr(li, k) for k, _ in li._fields_}
        d["filter_used"] = FILTER_NAMES.get(d["filter_used"], str(d["filter_used"]))
        return d


def pack_reads(seqs) -> tuple:
    """List of str/bytes -> (uint8 bytes, int64 offsets): the packed-chunk layout of the ABI
    (what the Julia shim builds from Chunk.data.seqs, core.jl:5-10)."""
    raw = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs]
    off = np.zeros(len(raw) + 1, dtype=np.int64)
    if raw:
        off[1:] = np.cumsum([len(b) for b in raw])
    blob = np.frombuffer(b"".join(raw), dtype=np.uint8).copy() if raw else np.zeros(0, dtype=np.uint8)
    return blob, off
