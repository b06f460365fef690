"""CPU: the fuzz generator + oracle under the whole option space (thread-count invariance of
the batch driver == the reference's 4-thread rerun, runtests.jl:30-35)."""
import numpy as np
import pytest

import fuzz
import helpers as H


@pytest.mark.parametrize("seed", range(24))
def test_oracle_thread_invariance(seed):
    cfg, seq, off = fuzz.random_case(1000 + seed, n_reads=300)
    a = H.orc.OracleClassifier(cfg, nthreads=1)
    b = H.orc.OracleClassifier(cfg, nthreads=4)
    fuzz.assert_same(a.classify(seq, off), b.classify(seq, off), f"seed {seed}")
    assert np.array_equal(a.counts, b.counts)
    assert a.counts[0] == len(off) - 1
    assert a.counts[1] + a.counts[2] + a.counts[3] == a.counts[0]
    assert a.counts[4:].sum() == a.counts[1]
