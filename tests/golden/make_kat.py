"""Writes tests/golden/kat.json: the known-answer vectors of the reference's own unit tests,
transcribed BY HAND from /root/reference/test/unit/*.jl (inputs and expected outputs only —
data, not source).  No reference code is executed (there is no Julia runtime in the build
container); this script only serialises the table below.  "Inf" encodes Float64 Inf.

Each vector: fn = the reference function under test, args = its positional arguments in the
reference's order (ranges as [first, last], `nothing` as null), expect = the asserted value,
src = file:line of the @test in the reference.
"""
import json
import os

INF = "Inf"
K = []


def add(fn, args, expect, src):
    K.append({"fn": fn, "args": args, "expect": expect, "src": src})


# ---- test/unit/alignment.jl ----
# semiglobal_alignment_N(ws, query, ref, max_error, match, mismatch, indel, nindel, range, max_start, min_end, non_N_m)
add("semiglobal_alignment_N", ["ANNC", "ATTC", 0.5, 0, 1, 1, 1, [1, 4], 1, 4, 2], 0.0, "unit/alignment.jl:27-28")
add("semiglobal_alignment_N", ["ANNC", "ATTG", 0.5, 0, 1, 1, 1, [1, 4], 1, 4, 2], 0.5, "unit/alignment.jl:35-36")
add("semiglobal_alignment_N", ["ANNC", "ATT", 0.5, 0, 1, 1, 1, [1, 3], 1, 3, 2], 0.5, "unit/alignment.jl:45-46")
# parse_dynamic_range -> [start_offset, start_from_end, end_offset, end_from_end]
add("parse_dynamic_range", ["1:10"], [1, False, 10, False], "unit/alignment.jl:51-55")
add("parse_dynamic_range", ["1:end"], [1, False, 0, True], "unit/alignment.jl:57-61")
add("parse_dynamic_range", ["end-5:end"], [-5, True, 0, True], "unit/alignment.jl:63-67")
# resolve(range_str, len) -> [first, last]
add("resolve", ["1:10", 100], [1, 10], "unit/alignment.jl:70-72")
add("resolve", ["1:end", 100], [1, 100], "unit/alignment.jl:74-75")
add("resolve", ["end-5:end", 100], [95, 100], "unit/alignment.jl:77-78")
# semiglobal_alignment(ws, query, ref, max_error, match, mismatch, indel, range, max_start, min_end[, trim_side[, need_tb]])
add("semiglobal_alignment", ["AAAA", "TTTTAAAA", 0.0, 0, 1, 1, [1, 8], 1, 8], INF, "unit/alignment.jl:101-102")

# ---- test/unit/trimming.jl ----
add("semiglobal_alignment", ["TTTTT", "AAAAATTTTTCCCCC", 0.0, 0, 1, 1, [1, 15], 100, 1, 3], [0.0, 6, 10],
    "unit/trimming.jl:19-22")
add("semiglobal_alignment", ["TTTTT", "AAAAATTTTTCCCCC", 0.0, 0, 1, 1, [1, 15], 100, 1, 5], [0.0, 6, 10],
    "unit/trimming.jl:49-52")
add("semiglobal_alignment", ["ACGT", "ACGTACGT", 0.0, 0, 1, 1, [1, 8], 100, 1, 3], [0.0, 5, 8],
    "unit/trimming.jl:82-85")
add("semiglobal_alignment", ["AA", "AAAA", 0.0, 0, 1, 1, [1, 4], 100, 1, 3], [0.0, 3, 4], "unit/trimming.jl:90-93")
add("semiglobal_alignment", ["TTTTT", "AAAAATTTTTCCCCC", 0.0, 0, 1, 1, [1, 15], 100, 1], 0.0,
    "unit/trimming.jl:140-142")
# determine_filename(read, config) with config given as DemuxConfig keyword overrides
add("determine_filename",
    ["AAAAATTTTTCCCCC", {"bc_seqs": ["TTTTT"], "bc_lengths_no_N": [5], "ids": ["id1"], "trim_side": 3}],
    ["id1.fastq", 1, 5], "unit/trimming.jl:33-37")
add("determine_filename",
    ["AAAAATTTTTCCCCC", {"bc_seqs": ["TTTTT"], "bc_lengths_no_N": [5], "ids": ["id1"], "trim_side": 5}],
    ["id1.fastq", 11, 15], "unit/trimming.jl:62-66")
add("determine_filename",
    ["AAAAATTTTTCCCCCGGGGGTTTTT",
     {"bc_seqs": ["TTTTT"], "bc_lengths_no_N": [5], "ids": ["id1"], "is_dual": True, "bc_seqs2": ["GGGGG"],
      "bc_lengths_no_N2": [5], "ids2": ["id2"], "trim_side": 5, "trim_side2": 3}],
    ["id1.id2.fastq", 11, 15], "unit/trimming.jl:125-129")

# ---- test/unit/hamming.jl ----
# hamming_align(query, ref, max_error_rate, range, max_start_pos, min_end_pos, trim_side)
add("hamming_align", ["AAAA", "TTAAAAgg", 0.2, [1, 8], 8, 1, None], [0.0, 3, 6], "unit/hamming.jl:10-11")
add("hamming_align", ["AAAA", "TTAATAgg", 0.3, [1, 8], 8, 1, None], [0.25, 3, 6], "unit/hamming.jl:16-17")
add("hamming_align", ["AAAA", "TTAATAgg", 0.2, [1, 8], 8, 1, None], [INF, -1, -1], "unit/hamming.jl:20-21")
add("hamming_align", ["ANNA", "TTAATAgg", 0.0, [1, 8], 8, 1, None], [0.0, 3, 6], "unit/hamming.jl:26-27")
add("hamming_align", ["AAAA", "TTANAAgg", 0.0, [1, 8], 8, 1, None], [INF, -1, -1], "unit/hamming.jl:35-36")
add("hamming_align", ["AAAA", "AAAA", 0.0, [1, 4], 4, 1, None], [0.0, 1, 4], "unit/hamming.jl:41-42")
add("hamming_align", ["AA", "AATAA", 0.0, [1, 5], 5, 1, 3], [0.0, 4, 5], "unit/hamming.jl:50-51")
add("hamming_align", ["AA", "AATAA", 0.0, [1, 5], 5, 1, None], [0.0, 1, 2], "unit/hamming.jl:54-55")

# ---- test/unit/exact.jl ----
# exact_align(query, ref, range, max_start_pos, min_end_pos, trim_side)
add("exact_align", ["AAAA", "TTAAAAgg", [1, 8], 8, 1, None], [0.0, 3, 6], "unit/exact.jl:9-10")
add("exact_align", ["AAAA", "TTAATAgg", [1, 8], 8, 1, None], [INF, -1, -1], "unit/exact.jl:15-16")
add("exact_align", ["ANNA", "TTAATAgg", [1, 8], 8, 1, None], [INF, -1, -1], "unit/exact.jl:23-24")
add("exact_align", ["AAAA", "TTANAAgg", [1, 8], 8, 1, None], [INF, -1, -1], "unit/exact.jl:29-30")
add("exact_align", ["AAAA", "AAAA", [1, 4], 4, 1, None], [0.0, 1, 4], "unit/exact.jl:35-36")
add("exact_align", ["AA", "AATAA", [1, 5], 5, 1, 3], [0.0, 4, 5], "unit/exact.jl:41-42")
add("exact_align", ["AA", "AATAA", [1, 5], 5, 1, 5], [0.0, 1, 2], "unit/exact.jl:45-46")
add("exact_align", ["AA", "AATAA", [1, 5], 5, 4, 3], [0.0, 4, 5], "unit/exact.jl:54-55")
add("exact_align", ["AA", "AATAA", [1, 5], 5, 6, 3], [INF, -1, -1], "unit/exact.jl:59-60")
add("exact_align", ["AA", "AATAA", [1, 5], 5, 3, 5], [0.0, 4, 5], "unit/exact.jl:83-84")

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat.json")
    with open(out, "w") as f:
        json.dump(K, f, indent=1)
    print(f"{len(K)} vectors -> {out}")
