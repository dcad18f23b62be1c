"""Kaldi text-file readers (interface of the reference's processing/readfiles.py)."""
import gzip
from collections import OrderedDict

import numpy as np


def _records(path, opener=open, mode="r"):
    """(raw line, space-separated fields) of every line, end-of-line removed"""
    with opener(path, mode) as handle:
        for raw in handle:
            yield raw, raw.rstrip("\n").split(" ")


def read_alignments(filename):
    """gzipped "<utt> <pdf> <pdf> ..." -> {utt: int array} (readfiles.py:9-28; a trailing blank is tolerated)"""
    return {rec[0]: np.asarray([int(tok) for tok in rec[1:] if tok != ""])
            for _, rec in _records(filename, gzip.open, "rt")}


def read_segments(filename):
    """kaldi `segments` lines "<segment> <recording> <begin> <end>" -> {recording: [(segment, begin, end), ...]},
    recordings and segments in file order (readfiles.py:30-57)"""
    table = OrderedDict()
    for _, (segment, recording, begin, end) in _records(filename):
        table.setdefault(recording, []).append((segment, float(begin), float(end)))
    return table


def read_wavfiles(filename):
    """kaldi `wav.scp` -> {utterance: (path, False)} for plain entries, {utterance: (command line, True)} for extended
    ones ("<utt> <command ...> |"), in file order (readfiles.py:59-87)"""
    table = OrderedDict()
    for raw, rec in _records(filename):
        if len(rec) == 2:
            table[rec[0]] = (rec[1], False)
        else:
            table[rec[0]] = (raw[len(rec[0]) + 1:len(raw) - 1], True)  # everything after the id, without the newline
    return table


def read_utt2spk(filename):
    """utt2spk "<utterance> <speaker>" lines -> dict (readfiles.py:89-105)"""
    return {rec[0]: rec[1] for _, rec in _records(filename)}
