"""Data ingest in the reference's file format: the step before the path.

Mirrors read-file->data / file->file-specs / create-walker-data of the reference
(mcmc-fitting.lisp:1425-1477, 827-830): delimiter-separated text (tab by default; `;` for the
NV files, nv-specific.lisp:10), Windows line ends allowed, header = every line before the first
one whose first field reads as a number, blank lines skipped, result transposed into one list
per COLUMN.  `example-data.xls` of the reference is this format (tab-separated text with CRLF,
1 header line, 9 columns).  Lisp users keep the reference's own reader; this one serves the
Python mirror.
"""
import re

import numpy as np


def _numberp(tok):
    """what (numberp (read-from-string tok)) accepts for the data files: 2.000E+3, -4.172d-7, 12"""
    t = tok.strip()
    if not t:
        return False
    try:
        float(t.replace("d", "e").replace("D", "E"))
        return True
    except ValueError:
        return False


def _read_number(tok):
    return float(tok.strip().replace("d", "e").replace("D", "E"))


def file_to_file_specs(filename, delim="\t"):
    """(file->file-specs filename :delim delim) M:1436-1450 as a dict with the same keys"""
    num_lines, found_data, data_length, data_rows = 0, None, None, None
    with open(filename, "r", newline="") as f:
        for raw in f:
            line = raw.rstrip("\n").rstrip("\r")
            if line == "" and found_data is not None and data_rows is None:
                data_rows = num_lines - found_data           # first blank line after data: page size
            elif line == "":
                continue
            # (numberp (read-from-string (elt (split-string #\tab line) 0))): the Lisp reader
            # stops at whitespace and treats ';' as a comment, which is why ;-separated files
            # pass this test as well
            elif found_data is None and _numberp(re.split(r"[;\s]", line.split("\t")[0].strip(), 1)[0]):
                found_data = num_lines
                data_length = len(line.split(delim))
                num_lines += 1
            else:
                num_lines += 1
    if found_data is None:
        raise ValueError("%s: no line starts with a number" % filename)
    rows = data_rows if data_rows else num_lines - found_data
    return {"file-lines": num_lines, "header-lines": found_data, "data-length": data_length,
            "data-rows": rows,
            "num-pages": (num_lines - found_data) // data_rows if data_rows else 1}


def read_file_to_data(filename, file_specs=None, delim="\t", transpose=True, pages=False):
    """(read-file->data filename &key file-specs delim transpose pages) M:1452-1477: a list of
    columns (transpose t) of floats; with pages, a list of such lists."""
    specs = file_specs or file_to_file_specs(filename, delim)
    rows = []
    with open(filename, "r", newline="") as f:
        for _ in range(specs["header-lines"]):
            f.readline()
        for raw in f:
            line = raw.rstrip("\n").rstrip("\r")
            if line == "":
                continue
            rows.append([_read_number(t) for t in line.split(delim)])
    if transpose:
        ncol = len(rows[0]) if rows else 0
        data = [[r[c] for r in rows] for c in range(ncol)]     # transpose-matrix M:601-603
    else:
        data = rows
    if pages:
        n, per = specs["num-pages"], specs["data-rows"]
        return [[col[p * per:(p + 1) * per] for col in data] for p in range(n)]
    return data


def create_walker_data(data, *columns):
    """(create-walker-data data &rest columns) M:827-830: the chosen columns as vectors, e.g.
    (create-walker-data data 1 4) of test.lisp:15"""
    return [np.asarray(data[c], dtype=np.float64) for c in columns]
