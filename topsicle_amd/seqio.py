"""Minimal FASTA / FASTQ (.gz) reader and writer for the host side.

Replaces the Biopython calls of the reference (Bio.SeqIO.parse / SeqIO.write,
allsteps.py:145, main.py:84-86) -- Biopython is not a dependency of this package.  Record ids
follow Biopython: the first whitespace-delimited token of the header line.
"""
from __future__ import annotations

import gzip
import io
import logging
from dataclasses import dataclass


@dataclass
class Record:
    id: str
    description: str      # full header line without the leading '>' / '@'
    seq: str
    qual: str | None = None

    def __len__(self):
        return len(self.seq)


def _open_text(path: str):
    if path.endswith(".gz"):
        return io.TextIOWrapper(gzip.open(path, "rb"), encoding="utf-8", newline="")
    return open(path, "rt", encoding="utf-8", newline="")


def check_file_type(filepath: str):
    """'fastq' / 'fasta' from the first character of the file, 0 if unknown (allsteps.py:36-50)."""
    try:
        with _open_text(filepath) as h:
            first = h.readline().strip()
    except Exception as e:  # same contract as the reference: log and return 0
        logging.error(f"Error checking file type: {e}")
        return 0
    if first.startswith("@"):
        return "fastq"
    if first.startswith(">"):
        return "fasta"
    logging.warning("Format cannot be identified. Check the input.")
    return 0


def _first_token(desc: str) -> str:
    parts = desc.split()
    return parts[0] if parts else ""


def parse(handle, fmt: str):
    if fmt == "fasta":
        desc, chunks = None, []
        for line in handle:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if desc is not None:
                    yield Record(_first_token(desc), desc, "".join(chunks))
                desc, chunks = line[1:], []
            elif desc is not None:
                chunks.append(line.strip())
        if desc is not None:
            yield Record(_first_token(desc), desc, "".join(chunks))
    elif fmt == "fastq":
        while True:
            head = handle.readline()
            if not head:
                return
            head = head.rstrip("\r\n")
            if not head:
                continue
            if not head.startswith("@"):
                raise ValueError("FASTQ record does not start with '@'")
            seq_lines = []
            line = handle.readline()
            while line and not line.startswith("+"):
                seq_lines.append(line.strip())
                line = handle.readline()
            seq = "".join(seq_lines)
            qual = ""
            while len(qual) < len(seq):
                q = handle.readline()
                if not q:
                    break
                qual += q.rstrip("\r\n")
            desc = head[1:]
            yield Record(_first_token(desc), desc, seq, qual)
    else:
        raise ValueError(f"unknown format {fmt!r}")


def read_records(filepath: str):
    """Generator over the records of a FASTA/FASTQ(.gz) file (unzip_file, allsteps.py:127-149).
    Parse errors are logged and end the iteration, as upstream."""
    if not isinstance(filepath, str):
        logging.error("Input must be a string representing the file path.")
        return
    fmt = check_file_type(filepath)
    if not fmt:
        logging.error("File type could not be determined or is unsupported.")
        return
    try:
        with _open_text(filepath) as h:
            yield from parse(h, fmt)
    except Exception as e:
        logging.error(f"Error parsing file: {e}")


def write_record(handle, rec: Record, fmt: str):
    """One record in the layout Biopython's SeqIO.write produces (FASTA wrapped at 60)."""
    if fmt == "fastq":
        if rec.qual is None:
            raise ValueError("no qualities for FASTQ output")
        handle.write(f"@{rec.description}\n{rec.seq}\n+\n{rec.qual}\n")
    elif fmt == "fasta":
        handle.write(f">{rec.description}\n")
        for i in range(0, len(rec.seq), 60):
            handle.write(rec.seq[i:i + 60] + "\n")
    else:
        raise ValueError(fmt)
