"""Evidence for the 'known-score class' used by the fused HIP kernel:
under SimpleScoring unit costs, ScoreOnly output and non-binding ranges, the reference's
banded cut-off DP (restated line by line in the oracle) returns exactly the plain full-matrix
semi-global distance d when d <= floor(rate*m), else Inf.  Differential test on random
(barcode, read, rate) triples incl. planted/mutated copies, N bases, empty reads, m = 1..32."""
import ctypes as C

import helpers as H


def test_core_equals_plain_dp_in_class():
    fb = (C.c_int64 * 6)()
    bad = H.orc.lib().orc_selftest_known_class(20260515, 400_000, fb)
    assert bad == 0, f"first disagreement (iter, m, n, core, plain, ae): {list(fb)}"


def test_unit_distance_sanity():
    u8 = lambda b: (C.c_uint8 * len(b)).from_buffer_copy(b)
    d = H.orc.lib().orc_unit_distance
    assert d(u8(b"ACGT"), 4, u8(b"TTACGTTT"), 8) == 0
    assert d(u8(b"ACGT"), 4, u8(b"TTACTTT"), 7) == 1
    assert d(u8(b"AAAA"), 4, u8(b"CCCC"), 4) == 4


def test_restricted_exact_run_equals_full_run():
    """running the reference's column loop only over e_lo - 2(m+kb) - 1 .. e_hi
    ([e_lo, e_hi] = first/last column with unit distance <= kb) returns the same (score, start, end)
    as the full run — weighted costs, N-scoring, ScoreOnly and traceback, every trim side, column
    windows, binding start/end ranges, tightened thresholds, repeated occurrences."""
    fb = (C.c_int64 * 8)()
    bad = H.orc.lib().orc_selftest_windowed_exact(20260515, 300_000, fb)
    assert bad == 0, f"first disagreement (iter, m, n, full.raw, res.raw, full.start, res.start, mode*10+trim): {list(fb)}"


def test_register_dp_formulation_equals_the_faithful_core():
    """The HIP register DP visits ALL rows and predicates state changes on fact..lact (csrc/bdx_core.h
    sg_core_reg); oracle/bdx_oracle.c kernel_model_core restates that formulation in C.  It must return the
    same (score, start, end) as the line-faithful core — unrestricted, and over the restricted column range
    of §3.2 together with the reachability cone (rows that cannot reach row m by the last column within the
    operation budget are skipped).  The cone is exact (0 disagreements in 6 M cases) but was not kept in the
    kernel: the union over the 64 lanes of a wave leaves almost every row block active."""
    fb = (C.c_int64 * 8)()
    bad = H.orc.lib().orc_selftest_cone(20260515, 300_000, fb)
    assert bad == 0, f"first disagreement (iter, m, n, full.raw, model.raw, full.start, model.start, 1000*plain + mode*10+trim): {list(fb)}"


def test_clean_class_full_dp_equals_the_cut_off_loop():
    """with in-domain SimpleScoring costs and start / end ranges that do not bind, a plain DP over
    all m rows records what the reference's banded cut-off loop records — values, origins under every tie rule,
    early exits, restricted column ranges, tightened thresholds (the model is csrc/bdx_core.h sg_core_clean)."""
    fb = (C.c_int64 * 8)()
    bad = H.orc.lib().orc_selftest_clean_class(20260515, 400_000, fb)
    assert bad == 0, list(fb)


def test_clean_class_short_lookback_for_score_and_end():
    """Score-only and end-only passes of the clean class: a restricted run that starts m + kb columns before the first
    end column (instead of 2 (m + kb) + 1) returns the reference's score and end column — also with a capped budget."""
    fb = (C.c_int64 * 8)()
    bad = H.orc.lib().orc_selftest_clean_short_lookback(20260515, 400_000, fb)
    assert bad == 0, list(fb)


def test_diagonal_band_dp_records_what_the_reference_records():
    """The exact stage's diagonal-band DP (csrc/bdx_core.h sg_core_band, oracle model band_dp): with the end columns of
    the tracked sweep known, a DP over the H >= W + 2 kb diagonals an alignment within the budget can touch returns
    the reference's (score, start, end) — all three output forms, both trim sides, ties, copies hanging over the
    read's ends, column windows that start inside the read, capped budgets (then: never a value within the cap that
    the reference does not return)."""
    fb = (C.c_int64 * 8)()
    bad = H.orc.lib().orc_selftest_band_class(20260515, 600_000, fb)
    assert bad == 0, list(fb)
    assert fb[1] > 50_000 and fb[2] > 20_000 and fb[3] > 5_000, list(fb)  # compared exactly / with traceback / band across the first column


def test_known_trim_class_reversed_sweep_gives_the_reference_start():
    """in the known-score class a trim_side = 3 pass reports the LARGEST origin among the alignments of the
    best score, and a right-to-left bit-vector sweep with the reversed barcode delivers it (last column that lowers the
    running minimum + the top bit of Eq & Pv there); trim_side = 5: the first column of a left-to-right sweep that attains
    the minimum.  The 32-bit model of the wave kernel's sweeps (oracle orc_known_trim_positions) against the line-faithful
    core: random and low-complexity pairs (ties), copies at the window's edges, column windows that start inside the read."""
    fb = (C.c_int64 * 8)()
    bad = H.orc.lib().orc_selftest_known_start(20260515, 600_000, fb)
    assert bad == 0, f"first disagreement (iter, m, n, core.raw, model d, core.start, model pos | core.end, trim | model pos): {list(fb)}"
    assert fb[1] > 50_000 and fb[2] > 500 and fb[3] > 20_000, list(fb)  # starts inside the read / starts <= 0 / ends compared


def test_known_alignment_class_anchored_sweeps_give_the_other_position():
    """with one position of a pass's winner known (its end for trim_side 5 / none, its start for trim_side 3),
    the other one comes out of an ANCHORED bit-vector sweep — right to left from the end with row 0 not free (the first column
    whose score equals the distance; + the diagonal-move bit), or left to right from a prepared "row 1 entered here" column (the
    first column whose score equals the distance).  The 32-bit model (oracle orc_known_other_position) behind the first sweep's
    model against the line-faithful core with traceback: every trim side, ties, window edges, starts <= 0 (handed on)."""
    fb = (C.c_int64 * 8)()
    bad = H.orc.lib().orc_selftest_known_alignment(20260515, 600_000, fb)
    assert bad == 0, f"first disagreement (iter, m, n, core.raw, core.start, core.end, known position, other * 10 + trim): {list(fb)}"
    assert fb[1] > 100_000 and fb[2] > 50_000 and fb[3] > 2_000, list(fb)  # starts from ends / ends from starts / starts <= 0
