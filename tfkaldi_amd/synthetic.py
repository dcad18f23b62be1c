"""Deterministic synthetic Kaldi data: real ark / scp / utt2spk / gzipped alignment
files, so benchmarks and tests go through the same feature + alignment I/O path as real data."""
import gzip
import os

import numpy as np

from .processing import ark


def write_text_targets(directory, num_utt, words_per_utt=(1, 4), seed=99):
    """text.txt with "<utt> WORD WORD ..." lines (upper-case words over a small vocabulary) for the utterance ids
    of write_corpus; returns the path"""
    rng = np.random.default_rng(seed)
    vocab = ["THE", "CAT", "SAT", "ON", "A", "MAT", "DOG", "RAN", ",COMMA", ".PERIOD", "IT'S", "<NOISE>"]
    path = os.path.join(directory, "text.txt")
    with open(path, "w") as fid:
        for i in range(num_utt):
            n = int(rng.integers(words_per_utt[0], words_per_utt[1] + 1))
            fid.write("utt%06d %s\n" % (i, " ".join(vocab[int(k)] for k in rng.integers(0, len(vocab), size=n))))
    return path


def write_corpus(directory, num_utt, num_pdfs, feat_dim=40, utt_len=64, num_speakers=4, feat_seed=1234,
                 ali_seed=4321, lengths=None):
    """feats.scp/.ark (float32 N(0,1)*2+5), cmvn.scp/.ark (per-speaker stats), utt2spk, maxlength and
    pdf.all (gz text "<utt> id id ...", ids uniform in [0, num_pdfs)).  Returns a dict of paths."""
    os.makedirs(directory, exist_ok=True)
    rng = np.random.default_rng(feat_seed)
    ali_rng = np.random.default_rng(ali_seed)
    paths = {k: os.path.join(directory, v) for k, v in dict(
        feats_scp="feats.scp", feats_ark="feats.ark", cmvn_scp="cmvn.scp", cmvn_ark="cmvn.ark",
        utt2spk="utt2spk", alignments="pdf.all", maxlength="maxlength").items()}
    for k in ("feats_ark", "cmvn_ark"):
        if os.path.exists(paths[k]):
            os.remove(paths[k])
    writer = ark.ArkWriter(paths["feats_scp"], paths["feats_ark"])
    stats = {}
    maxlen = 0
    with open(paths["utt2spk"], "w") as u2s, gzip.open(paths["alignments"], "wt") as ali:
        for i in range(num_utt):
            utt, spk = "utt%06d" % i, "spk%02d" % (i % num_speakers)
            n = int(lengths[i]) if lengths is not None else utt_len
            maxlen = max(maxlen, n)
            feats = (rng.standard_normal((n, feat_dim)) * 2 + 5).astype(np.float32)
            writer.write_next_utt(utt, feats)
            st = stats.setdefault(spk, np.zeros((2, feat_dim + 1), dtype=np.float64))
            st[0, :feat_dim] += feats.sum(0)
            st[0, feat_dim] += n
            st[1, :feat_dim] += (feats.astype(np.float64) ** 2).sum(0)
            u2s.write("%s %s\n" % (utt, spk))
            ali.write("%s %s\n" % (utt, " ".join(str(x) for x in ali_rng.integers(0, num_pdfs, size=n))))
    writer.close()
    cmvn = ark.ArkWriter(paths["cmvn_scp"], paths["cmvn_ark"])
    for spk in sorted(stats):
        cmvn.write_next_utt(spk, stats[spk])
    cmvn.close()
    with open(paths["maxlength"], "w") as fid:
        fid.write(str(maxlen))
    return paths
