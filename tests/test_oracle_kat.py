"""Pins the CPU oracle (and the host-side range DSL) to the reference's own known-answer
tests: every @test of /root/reference/test/unit/{alignment,trimming,hamming,exact}.jl,
transcribed in tests/golden/kat.json (see tests/golden/make_kat.py)."""
import pytest

from helpers import kat_vectors, run_kat

KAT = kat_vectors()


@pytest.mark.parametrize("vec", KAT, ids=[f"{v['fn']}@{v['src']}" for v in KAT])
def test_kat_oracle(vec):
    got, expect = run_kat(vec, "oracle")
    assert got == expect, vec["src"]


def test_kat_count():
    # 3 + 6 + 1 (alignment.jl) + 5 + 3 (trimming.jl) + 8 (hamming.jl) + 10 (exact.jl)
    assert len(KAT) == 36
