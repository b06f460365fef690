"""Barcode-file preprocessing and FASTQ open helpers (host side).

Mirrors BioDemuX.jl src/fileio.jl:7-72 (preprocess_bc_file) and :77-113 (smart_open,
read_fastq, write_fastq).  These sit either side of the hot path; they are kept minimal and
exist so the golden-file tests can drive the C-ABI through the reference's file contract.
"""
from __future__ import annotations

import csv
import gzip
import re
from contextlib import contextmanager
from typing import List, Tuple

_COMPLEMENT = {  # fileio.jl:58-62
    "A": "T", "T": "A", "G": "C", "C": "G",
    "a": "t", "t": "a", "g": "c", "c": "g",
    "N": "N", "n": "n",
}


def preprocess_bc_file(bc_file: str, complement: bool, rev: bool) -> Tuple[List[str], List[int], List[str]]:
    """fileio.jl:7-72.  Returns ``(sequences, lengths_no_N, ids)``."""
    sequences: List[str] = []
    ids: List[str] = []
    low = bc_file.lower()
    if low.endswith(".fasta") or low.endswith(".fa"):  # fileio.jl:11-32
        current_seq = ""
        with open(bc_file, "r") as io:
            for line in io:
                line = line.rstrip("\n").rstrip("\r") if line.endswith("\n") else line
                if line.startswith(">"):
                    if current_seq:
                        sequences.append(current_seq)
                        current_seq = ""
                    current_id = re.sub(r"\s.*$", "", line[1:].strip(), flags=re.S)
                    ids.append(current_id)
                else:
                    current_seq += line.strip()
            if current_seq:
                sequences.append(current_seq)
        annotations = ["B" * len(seq) for seq in sequences]
    else:  # fileio.jl:34-40
        delim = "," if low.endswith(".csv") else "\t"
        with open(bc_file, "r", newline="") as io:
            rows = list(csv.DictReader(io, delimiter=delim))
        try:
            sequences = [str(r["Full_seq"]) for r in rows]
            ids = [str(r["ID"]) for r in rows]
            annotations = [str(r["Full_annotation"]) for r in rows]
        except KeyError as e:
            raise KeyError(f"barcode file {bc_file} lacks column {e}") from None

    for i in range(len(sequences)):  # fileio.jl:44-50
        if len(sequences[i]) != len(annotations[i]):
            raise ValueError(f"Length mismatch between sequence and annotation for ID: {ids[i]}")
        sequences[i] = "".join(c for c, a in zip(sequences[i], annotations[i]) if a == "B")

    sequences = [s.upper() for s in sequences]  # fileio.jl:54
    sequences = [s.replace("U", "T") for s in sequences]  # fileio.jl:55
    if complement:  # fileio.jl:57-64
        sequences = ["".join(_COMPLEMENT.get(c, c) for c in s) for s in sequences]
    if rev:  # fileio.jl:65-67
        sequences = [s[::-1] for s in sequences]
    bc_lengths_no_N = [sum(1 for c in s if c != "N") for s in sequences]  # fileio.jl:69
    return sequences, bc_lengths_no_N, ids


@contextmanager
def smart_open(filepath: str, mode: str):
    """fileio.jl:77-95: gzip iff the path ends with .gz (case-insensitive).  Binary streams."""
    is_gzip = filepath.lower().endswith(".gz")
    bmode = {"r": "rb", "w": "wb", "a": "ab"}[mode]
    f = gzip.open(filepath, bmode) if is_gzip else open(filepath, bmode)
    try:
        yield f
    finally:
        f.close()


def read_fastq(filepath: str):
    """fileio.jl:102-104."""
    return smart_open(filepath, "r")


def write_fastq(filepath: str):
    """fileio.jl:111-113 (append mode)."""
    return smart_open(filepath, "a")
