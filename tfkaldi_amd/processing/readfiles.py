"""Kaldi text-file readers (reference processing/readfiles.py)."""
import gzip
from collections import OrderedDict

import numpy as np


def read_alignments(filename):
    """gzipped "<utt> <pdf> <pdf> ..." lines -> {utt: int array} (readfiles.py:9-28)"""
    alignments = {}
    with gzip.open(filename, "rt") as fid:
        for line in fid:
            data = line.replace(" \n", "").replace("\n", "").split(" ")
            alignments[data[0]] = np.asarray([int(x) for x in data[1:]])
    return alignments


def read_segments(filename):
    """kaldi `segments`: "<segment> <recording> <begin> <end>" -> {recording: [(segment, begin, end), ...]} in file
    order (readfiles.py:30-57)"""
    segments = OrderedDict()
    with open(filename) as fid:
        for line in fid:
            data = line.replace("\n", "").split(" ")
            segments.setdefault(data[1], []).append((data[0], float(data[2]), float(data[3])))
    return segments


def read_wavfiles(filename):
    """kaldi `wav.scp` -> {utterance: (filename, False) | (command line, True)} in file order (readfiles.py:59-87)"""
    wavfiles = OrderedDict()
    with open(filename) as fid:
        for line in fid:
            data = line.replace("\n", "").split(" ")
            if len(data) == 2:
                wavfiles[data[0]] = (data[1], False)
            else:  # an extended filename: a command that writes the wav to its standard output
                wavfiles[data[0]] = (line[len(data[0]) + 1:len(line) - 1], True)
    return wavfiles


def read_utt2spk(filename):
    """utt2spk: one "<utterance> <speaker>" pair per line -> dict (reference readfiles.py:89-105)."""
    utt2spk = {}
    with open(filename) as fid:
        for line in fid:
            fields = line.replace("\n", "").split(" ")
            utt2spk[fields[0]] = fields[1]
    return utt2spk
