"""Range DSL of the reference (``"1:end"``, ``"end-5:end"`` ...).

Mirrors BioDemuX.jl src/classification.jl:9-14 (DynamicRange), :61-81 (parse_part),
:83-94 (parse_dynamic_range) and :96-100 (resolve).  Host-side only: the device receives
the four integers of a DynamicRange and resolves them per read.
"""
from __future__ import annotations

import re
from dataclasses import dataclass

_INT = re.compile(r"^[+-]?[0-9]+$")


def _parse_int(s: str) -> int:
    s = s.strip()
    if not _INT.match(s):
        raise ValueError(f"cannot parse {s!r} as an integer")
    return int(s)


@dataclass(frozen=True)
class DynamicRange:
    """classification.jl:9-14."""

    start_offset: int
    start_from_end: bool
    end_offset: int
    end_from_end: bool


def parse_part(s: str):
    """classification.jl:61-81: one side of ``a:b``; ``end`` becomes 0 + from_end flag;
    a single ``-`` or ``+`` is evaluated on the first two operands."""
    s = s.strip()
    from_end = "end" in s
    if from_end:
        s = s.replace("end", "0")
    if "-" in s:
        p = s.split("-")
        val = _parse_int(p[0]) - _parse_int(p[1])
    elif "+" in s:
        p = s.split("+")
        val = _parse_int(p[0]) + _parse_int(p[1])
    else:
        val = _parse_int(s)
    return val, from_end


def parse_dynamic_range(range_str: str) -> DynamicRange:
    """classification.jl:83-94."""
    parts = range_str.split(":")
    if len(parts) != 2:
        raise ValueError(f"Invalid range format: {range_str}. Expected 'start:end'.")
    start_offset, start_from_end = parse_part(parts[0])
    end_offset, end_from_end = parse_part(parts[1])
    return DynamicRange(start_offset, start_from_end, end_offset, end_from_end)


def resolve(dr: DynamicRange, length: int):
    """classification.jl:96-100.  Returns ``(first, last)`` of ``max(1,s):min(len,e)`` with
    Julia's UnitRange normalisation: an empty range has ``last == first - 1``."""
    s = length + dr.start_offset if dr.start_from_end else dr.start_offset
    e = length + dr.end_offset if dr.end_from_end else dr.end_offset
    a = max(1, s)
    b = min(length, e)
    if b < a:
        b = a - 1
    return a, b
