"""merge_stats through the C-ABI's RCCL communicator (bdx_comm_*, bdx_allreduce_counts*): the forms a
1-GPU box can run — a one-rank ncclCommInitAll, a one-rank ncclCommInitRank from a unique id, and the
communicator-less copy — must all deliver the context's own counter vector, leave the per-rank counters
untouched and keep working across further batches."""
import numpy as np
import pytest

import helpers as H
from biodemux_jl_amd import hipabi, synth

pytestmark = pytest.mark.gpu


def _case():
    bcs = synth.make_barcodes(16, 16, seed=9, min_hamming=5)
    seq, off, _ = synth.make_reads(bcs, 20000, 80, seed=9)
    cfg = H.bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[16] * 16, ids=[str(i) for i in range(16)], max_error_rate=0.2)
    return cfg, seq, off


@pytest.mark.parametrize("mode", ["none", "init_all", "init_rank"])
def test_one_rank_allreduce(mode):
    cfg, seq, off = _case()
    with H.bdx.HipClassifier(cfg) as hc:
        if mode == "init_all":
            hipabi.comm_init_all([hc])
        elif mode == "init_rank":
            uid = hipabi.comm_unique_id()
            assert len(uid) == hipabi.BDX_COMM_ID_BYTES and any(uid)
            hc.comm_init_rank(uid, 0, 1)
        assert hc.comm_size == 1 and hc.comm_rank == 0
        hc.classify(seq, off)
        if mode == "init_all":
            hipabi.allreduce_counts_all([hc])
        else:
            hc.allreduce_counts()
        c1 = hc.counts
        assert np.array_equal(hc.reduced_counts, c1) and c1[0] == 20000
        hc.classify(seq, off)  # accumulation goes on; the reduced vector is a snapshot until the next all-reduce
        assert np.array_equal(hc.reduced_counts, c1)
        hc.allreduce_counts()
        assert np.array_equal(hc.reduced_counts, 2 * c1)
        if mode != "none":
            with pytest.raises(H.bdx.BdxError, match="already has a communicator"):
                hc.comm_init_rank(hipabi.comm_unique_id(), 0, 1)
            hc.comm_destroy()
            assert hc.comm_size == 1


def test_comm_argument_errors():
    cfg, _, _ = _case()
    with H.bdx.HipClassifier(cfg) as a, H.bdx.HipClassifier(cfg) as b:
        with pytest.raises(H.bdx.BdxError, match="share device"):
            hipabi.comm_init_all([a, b])  # RCCL wants one rank per device
        with pytest.raises(H.bdx.BdxError, match="bad communicator arguments"):
            a.comm_init_rank(bytes(128), 3, 2)
        with pytest.raises(H.bdx.BdxError, match="has not been called"):
            _ = a.reduced_counts
