import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is C: compile it once per session (gcc, seconds)."""
    import bdx_oracle

    bdx_oracle.build()


@pytest.fixture(scope="session", autouse=True)
def _poison_hand_over_buffers():
    """GPU tests run with BDX_POISON (read in bdx_create): before every classify call the library fills its hand-over
    buffers — candidate masks, window entries and counts, the hand-over lists, the output staging — with 0xA5, so a
    consumer that reads something no producer wrote gets garbage on EVERY run instead of whatever the allocator
    happened to hand back (the class of defect behind fuzz seed 53109).  BDX_TEST_NO_POISON=1 switches it off."""
    if os.environ.get("BDX_TEST_NO_POISON"):
        yield
        return
    old = os.environ.get("BDX_POISON")
    os.environ["BDX_POISON"] = "1"
    yield
    if old is None:
        os.environ.pop("BDX_POISON", None)
    else:
        os.environ["BDX_POISON"] = old


@pytest.fixture(autouse=True)
def _no_refused_hand_over_windows(request):
    """The exact kernel refuses a hand-over window that does not end inside the read and falls back to the whole pass
    window — results stay right, which would MASK a producer / consumer mismatch.  Every refusal is counted
    (bdx_rejected_windows); a GPU test during which the process-wide total moved fails."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from biodemux_jl_amd import hipabi

    lib = hipabi.load_library()
    before = int(lib.bdx_debug_rejected_windows_total())
    yield
    import gc

    gc.collect()  # contexts fold their counter into the total when they are destroyed
    after = int(lib.bdx_debug_rejected_windows_total())
    assert after == before, f"{after - before} hand-over violation(s) during this test: windows the exact kernel refused, or elements a consumer was about to read that no producer had written (BDX_POISON checker)"
