"""ctypes wrapper around oracle/libbdx_oracle.so — CPU ORACLE, test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (biodemux.jl_amd) never does.

The wrapper accepts any object carrying the reference's DemuxConfig field names
(classification.jl:16-58): max_error_rate, min_delta, match, mismatch, indel, nindel,
ref_search_range[2], barcode_start_range[2], barcode_end_range[2], bc_seqs[2],
bc_lengths_no_N[2], is_dual, trim_side[2], summary, matching_algorithm.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbdx_oracle.so")

ALG = {"semiglobal": 0, "hamming": 1, "exact": 2}
INF_INT = (2**63 - 1) // 4


class OrcRange(C.Structure):
    _fields_ = [
        ("start_offset", C.c_int64),
        ("start_from_end", C.c_int32),
        ("end_offset", C.c_int64),
        ("end_from_end", C.c_int32),
    ]


class OrcPass(C.Structure):
    _fields_ = [
        ("ref_search_range", OrcRange),
        ("barcode_start_range", OrcRange),
        ("barcode_end_range", OrcRange),
        ("trim_side", C.c_int32),
        ("n_barcodes", C.c_int32),
        ("bc_bytes", C.POINTER(C.c_uint8)),
        ("bc_off", C.POINTER(C.c_int64)),
        ("bc_len_no_N", C.POINTER(C.c_int64)),
    ]


class OrcConfig(C.Structure):
    _fields_ = [
        ("algorithm", C.c_int32),
        ("max_error_rate", C.c_double),
        ("min_delta", C.c_double),
        ("match", C.c_int64),
        ("mismatch", C.c_int64),
        ("indel", C.c_int64),
        ("has_nindel", C.c_int32),
        ("nindel", C.c_int64),
        ("is_dual", C.c_int32),
        ("summary", C.c_int32),
        ("pass_", OrcPass * 2),
    ]


class OrcAlign(C.Structure):
    _fields_ = [("score", C.c_double), ("raw", C.c_int64), ("start", C.c_int64), ("end", C.c_int64)]


class OrcBest(C.Structure):
    _fields_ = [
        ("bc", C.c_int64),
        ("score", C.c_double),
        ("delta", C.c_double),
        ("start", C.c_int64),
        ("end", C.c_int64),
    ]


class OrcVerdict(C.Structure):
    _fields_ = [
        ("bc1", C.c_int32),
        ("bc2", C.c_int32),
        ("keep_start", C.c_int32),
        ("keep_end", C.c_int32),
        ("pass_status", C.c_int32 * 2),
        ("pass_bc", C.c_int32 * 2),
        ("pass_start", C.c_int32 * 2),
        ("pass_end", C.c_int32 * 2),
        ("pass_score", C.c_double * 2),
        ("pass_delta", C.c_double * 2),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "bdx_oracle.c")
    hdr = os.path.join(_HERE, "bdx_oracle.h")
    stale = (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libbdx_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u8p = C.POINTER(C.c_uint8)
        i64p = C.POINTER(C.c_int64)
        i32p = C.POINTER(C.c_int32)
        f64p = C.POINTER(C.c_double)
        L.orc_semiglobal_alignment.restype = OrcAlign
        L.orc_semiglobal_alignment.argtypes = [
            u8p, C.c_int64, u8p, C.c_int64, C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
            C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
        ]
        L.orc_exact_align.restype = OrcAlign
        L.orc_exact_align.argtypes = [u8p, C.c_int64, u8p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32]
        L.orc_hamming_align.restype = OrcAlign
        L.orc_hamming_align.argtypes = [
            u8p, C.c_int64, u8p, C.c_int64, C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
        ]
        L.orc_find_best_matching_bc.restype = OrcBest
        L.orc_find_best_matching_bc.argtypes = [
            C.POINTER(OrcConfig), C.c_int, u8p, C.c_int64, i64p, i64p, C.c_int64, C.c_int64, C.c_int64,
            C.c_int64, C.c_int32, C.c_int32,
        ]
        L.orc_resolve.restype = None
        L.orc_resolve.argtypes = [C.POINTER(OrcRange), C.c_int64, i64p, i64p]
        L.orc_determine_filename.restype = None
        L.orc_determine_filename.argtypes = [C.POINTER(OrcConfig), u8p, C.c_int64, i64p, i64p, C.POINTER(OrcVerdict)]
        L.orc_classify_batch.restype = C.c_int
        L.orc_classify_batch.argtypes = [
            C.POINTER(OrcConfig), u8p, i64p, C.c_int64, i32p, i32p, i32p, i32p, i32p, i32p, f64p, i32p, f64p, i64p,
            C.c_int32,
        ]
        L.orc_selftest_known_class.restype = C.c_int64
        L.orc_selftest_known_class.argtypes = [C.c_uint64, C.c_int64, i64p]
        L.orc_selftest_windowed_exact.restype = C.c_int64
        L.orc_selftest_windowed_exact.argtypes = [C.c_uint64, C.c_int64, i64p]
        L.orc_selftest_cone.restype = C.c_int64
        L.orc_selftest_cone.argtypes = [C.c_uint64, C.c_int64, i64p]
        L.orc_selftest_clean_class.restype = C.c_int64
        L.orc_selftest_clean_class.argtypes = [C.c_uint64, C.c_int64, i64p]
        L.orc_selftest_clean_short_lookback.restype = C.c_int64
        L.orc_selftest_clean_short_lookback.argtypes = [C.c_uint64, C.c_int64, i64p]
        L.orc_selftest_band_class.restype = C.c_int64
        L.orc_selftest_band_class.argtypes = [C.c_uint64, C.c_int64, i64p]
        L.orc_selftest_known_start.restype = C.c_int64
        L.orc_selftest_known_start.argtypes = [C.c_uint64, C.c_int64, i64p]
        L.orc_selftest_known_alignment.restype = C.c_int64
        L.orc_selftest_known_alignment.argtypes = [C.c_uint64, C.c_int64, i64p]
        L.orc_unit_distance.restype = C.c_int64
        L.orc_unit_distance.argtypes = [u8p, C.c_int64, u8p, C.c_int64]
        _lib = L
    return _lib


def _u8(b: bytes):
    arr = (C.c_uint8 * max(1, len(b))).from_buffer_copy(b if len(b) else b"\0")
    return arr


def _ts(trim_side) -> int:
    return 0 if trim_side is None else int(trim_side)


def _alg(a) -> int:
    return ALG[str(a).lstrip(":")]


# ---- unit-level functions (signatures follow the reference's exported functions) ----


def semiglobal_alignment(query: str, ref: str, max_error: float, match: int, mismatch: int, indel: int,
                         ref_search_range, max_start_pos: int, min_end_pos: int, trim_side=None,
                         need_traceback: bool = False):
    """classification.jl:447.  ref_search_range = (first, last) inclusive, 1-based."""
    q, r = query.encode("latin-1"), ref.encode("latin-1")
    a = lib().orc_semiglobal_alignment(_u8(q), len(q), _u8(r), len(r), max_error, match, mismatch, indel, 0, 0,
                                       ref_search_range[0], ref_search_range[1], max_start_pos, min_end_pos,
                                       len(q), _ts(trim_side), int(need_traceback))
    if trim_side is None and not need_traceback:
        return a.score
    return (a.score, a.start, a.end)


def semiglobal_alignment_N(query: str, ref: str, max_error: float, match: int, mismatch: int, indel: int,
                           nindel: int, ref_search_range, max_start_pos: int, min_end_pos: int, non_N_m: int,
                           trim_side=None, need_traceback: bool = False):
    """classification.jl:463."""
    q, r = query.encode("latin-1"), ref.encode("latin-1")
    a = lib().orc_semiglobal_alignment(_u8(q), len(q), _u8(r), len(r), max_error, match, mismatch, indel, 1, nindel,
                                       ref_search_range[0], ref_search_range[1], max_start_pos, min_end_pos,
                                       non_N_m, _ts(trim_side), int(need_traceback))
    if trim_side is None and not need_traceback:
        return a.score
    return (a.score, a.start, a.end)


def exact_align(query: str, ref: str, ref_search_range, max_start_pos: int, min_end_pos: int, trim_side):
    """classification.jl:485."""
    q, r = query.encode("latin-1"), ref.encode("latin-1")
    a = lib().orc_exact_align(_u8(q), len(q), _u8(r), len(r), ref_search_range[0], ref_search_range[1],
                              max_start_pos, min_end_pos, _ts(trim_side))
    return (a.score, a.start, a.end)


def hamming_align(query: str, ref: str, max_error_rate: float, ref_search_range, max_start_pos: int,
                  min_end_pos: int, trim_side):
    """classification.jl:557."""
    q, r = query.encode("latin-1"), ref.encode("latin-1")
    a = lib().orc_hamming_align(_u8(q), len(q), _u8(r), len(r), max_error_rate, ref_search_range[0],
                                ref_search_range[1], max_start_pos, min_end_pos, _ts(trim_side))
    return (a.score, a.start, a.end)


# ---- config marshalling ----


def _mk_range(dr) -> OrcRange:
    return OrcRange(int(dr.start_offset), int(bool(dr.start_from_end)), int(dr.end_offset), int(bool(dr.end_from_end)))


class OracleConfig:
    """Owns the ctypes buffers of an orc_config_t built from a DemuxConfig-like object."""

    def __init__(self, cfg):
        self.src = cfg
        c = OrcConfig()
        c.algorithm = _alg(cfg.matching_algorithm)
        c.max_error_rate = float(cfg.max_error_rate)
        c.min_delta = float(cfg.min_delta)
        c.match, c.mismatch, c.indel = int(cfg.match), int(cfg.mismatch), int(cfg.indel)
        c.has_nindel = 0 if cfg.nindel is None else 1
        c.nindel = 0 if cfg.nindel is None else int(cfg.nindel)
        c.is_dual = int(bool(cfg.is_dual))
        c.summary = int(bool(cfg.summary))
        self._keep = []
        passes = [
            (cfg.ref_search_range, cfg.barcode_start_range, cfg.barcode_end_range, cfg.bc_seqs,
             cfg.bc_lengths_no_N, cfg.trim_side),
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
            off = np.zeros(len(raw) + 1, dtype=np.int64)
            if raw:
                off[1:] = np.cumsum([len(b) for b in raw])
            blob = np.frombuffer(b"".join(raw) + b"\0", dtype=np.uint8).copy()
            ln = np.asarray(list(lens) if len(lens) else [0], dtype=np.int64)
            self._keep += [blob, off, ln]
            P.bc_bytes = blob.ctypes.data_as(C.POINTER(C.c_uint8))
            P.bc_off = off.ctypes.data_as(C.POINTER(C.c_int64))
            P.bc_len_no_N = ln.ctypes.data_as(C.POINTER(C.c_int64))
        self.c = c
        self.B1 = len(cfg.bc_seqs)
        self.B2 = len(cfg.bc_seqs2) if cfg.is_dual else 0
        self.n_counts = 4 + self.B1 * max(1, self.B2)
        self.max_m = max([len(s) for s in cfg.bc_seqs] + [len(s) for s in (cfg.bc_seqs2 if cfg.is_dual else [])] + [1])


def determine_filename(seq: str, cfg):
    """classification.jl:871 — returns (bc1, bc2, keep_start, keep_end) indices, not the filename."""
    oc = cfg if isinstance(cfg, OracleConfig) else OracleConfig(cfg)
    s = seq.encode("latin-1") if isinstance(seq, str) else bytes(seq)
    DP = (C.c_int64 * (oc.max_m + 2))()
    origin = (C.c_int64 * (oc.max_m + 2))()
    v = OrcVerdict()
    lib().orc_determine_filename(C.byref(oc.c), _u8(s), len(s), DP, origin, C.byref(v))
    return v


def find_best_matching_bc(seq: str, cfg, pass_idx: int, ref_search_range, max_start_pos: int, min_end_pos: int,
                          trim_side, need_traceback: bool = False):
    """classification.jl:722 — returns (bc_idx, score, delta, start, end)."""
    oc = cfg if isinstance(cfg, OracleConfig) else OracleConfig(cfg)
    s = seq.encode("latin-1") if isinstance(seq, str) else bytes(seq)
    DP = (C.c_int64 * (oc.max_m + 2))()
    origin = (C.c_int64 * (oc.max_m + 2))()
    b = lib().orc_find_best_matching_bc(C.byref(oc.c), pass_idx, _u8(s), len(s), DP, origin, ref_search_range[0],
                                        ref_search_range[1], max_start_pos, min_end_pos, _ts(trim_side),
                                        int(need_traceback))
    return (b.bc, b.score, b.delta, b.start, b.end)


class OracleClassifier:
    """Batch classifier with the same duck-typed interface as the product's HipClassifier
    (classify(seq_bytes, seq_off) -> dict of numpy arrays; .counts).  Used by tests to check the
    host-side file contract on CPU and as the expected value in the GPU parity tests."""

    def __init__(self, cfg, nthreads: int = 1, want_pass: bool = True):
        self.oc = OracleConfig(cfg)
        self.nthreads = nthreads
        self.want_pass = want_pass
        self.counts = np.zeros(self.oc.n_counts, dtype=np.int64)

    def classify(self, seq_bytes: np.ndarray, seq_off: np.ndarray) -> dict:
        seq_bytes = np.ascontiguousarray(seq_bytes, dtype=np.uint8)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
        n = len(seq_off) - 1
        if seq_bytes.size == 0:
            seq_bytes = np.zeros(1, dtype=np.uint8)
        out = {
            "bc1": np.zeros(n, dtype=np.int32),
            "bc2": np.zeros(n, dtype=np.int32),
            "keep_start": np.zeros(n, dtype=np.int32),
            "keep_end": np.zeros(n, dtype=np.int32),
        }
        ps = pe = psc = pbc = pdl = None
        if self.want_pass:
            out["pass_start"] = np.zeros((n, 2), dtype=np.int32)
            out["pass_end"] = np.zeros((n, 2), dtype=np.int32)
            out["pass_score"] = np.zeros((n, 2), dtype=np.float64)
            out["pass_bc"] = np.zeros((n, 2), dtype=np.int32)
            out["pass_delta"] = np.zeros((n, 2), dtype=np.float64)
            ps = out["pass_start"].ctypes.data_as(C.POINTER(C.c_int32))
            pe = out["pass_end"].ctypes.data_as(C.POINTER(C.c_int32))
            psc = out["pass_score"].ctypes.data_as(C.POINTER(C.c_double))
            pbc = out["pass_bc"].ctypes.data_as(C.POINTER(C.c_int32))
            pdl = out["pass_delta"].ctypes.data_as(C.POINTER(C.c_double))
        i32 = C.POINTER(C.c_int32)
        rc = lib().orc_classify_batch(
            C.byref(self.oc.c), seq_bytes.ctypes.data_as(C.POINTER(C.c_uint8)),
            seq_off.ctypes.data_as(C.POINTER(C.c_int64)), n,
            out["bc1"].ctypes.data_as(i32), out["bc2"].ctypes.data_as(i32),
            out["keep_start"].ctypes.data_as(i32), out["keep_end"].ctypes.data_as(i32),
            ps, pe, psc, pbc, pdl, self.counts.ctypes.data_as(C.POINTER(C.c_int64)), self.nthreads)
        assert rc == 0
        return out

    def close(self):
        pass


def isinf(x) -> bool:
    return math.isinf(x)
