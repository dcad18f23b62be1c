"""Kaldi text-file readers used on the training path (reference processing/readfiles.py)."""


def read_utt2spk(filename):
    """utt2spk: one "<utterance> <speaker>" pair per line -> dict (reference readfiles.py:89-105)."""
    utt2spk = {}
    with open(filename) as fid:
        for line in fid:
            fields = line.replace("\n", "").split(" ")
            utt2spk[fields[0]] = fields[1]
    return utt2spk
