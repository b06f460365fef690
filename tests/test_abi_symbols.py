"""The C-ABI library must load on a GPU-less host and export every symbol that
include/biodemux_hip.h declares; without a device bdx_create must fail loudly (no CPU
fallback)."""
import ctypes
import os
import re

import pytest

import helpers as H
from biodemux_jl_amd import hipabi


def _declared_symbols():
    hdr = open(os.path.join(H.ROOT, "include", "biodemux_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(bdx_[a-z_0-9]+)\s*\(", hdr)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(hipabi.ABI_SYMBOLS)


def test_library_exports_every_symbol():
    assert os.path.exists(hipabi.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(hipabi.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.bdx_abi_version() == hipabi.BDX_ABI_VERSION


def test_struct_layout_matches_header():
    # sizes the C side checks in bdx_create (struct_size) — catches drift of the ctypes mirror
    assert ctypes.sizeof(hipabi.BdxRange) == 24
    assert ctypes.sizeof(hipabi.BdxPass) == 3 * 24 + 8 + 3 * 8 + 8 + 4 * 8
    assert ctypes.sizeof(hipabi.BdxOutputs) == 10 * 8


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = H.bdx.DemuxConfig(bc_seqs=["ACGT"], bc_lengths_no_N=[4], ids=["a"])
    with pytest.raises(H.bdx.BdxError, match="no HIP device|no CPU fallback|failed"):
        H.bdx.HipClassifier(cfg)


def test_config_validation_errors_before_device():
    lib = hipabi.load_library()
    cfg = H.bdx.DemuxConfig(bc_seqs=["ACGT"], bc_lengths_no_N=[4], ids=["a"], trim_side=4)
    c, keep = hipabi.pack_config(cfg)
    h = ctypes.c_void_p()
    assert lib.bdx_create(ctypes.byref(c), ctypes.byref(h)) == -1
    assert b"trim_side must be 3 or 5" in lib.bdx_last_error(None)
