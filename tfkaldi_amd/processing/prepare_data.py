"""Feature preparation with the reference's interface (processing/prepare_data.py): wav.scp (+ segments) -> feats.ark /
feats.scp / maxlength, per-speaker CMVN statistics, the shuffled scp -- with the arithmetic on the GPU, whole batches of
utterances per launch (csrc/features.hip).  File formats and file names are the reference's."""
import io
import os
import subprocess
from random import shuffle
from shutil import copyfile

import numpy as np
import scipy.io.wavfile as wav

from .. import features as device_features
from . import ark, feat, readfiles

# samples / feature bytes gathered on the host before one device pass
BATCH_SAMPLES = int(os.environ.get("TFK_FEAT_BATCH_SAMPLES", 1 << 26))
BATCH_CMVN_BYTES = int(os.environ.get("TFK_CMVN_BATCH_BYTES", 1 << 29))


def prepare_data(datadir, featdir, conf, feat_type, dynamic):
    """compute the features of all segments and save them on disk (prepare_data.py:13-78)

    The reference's `segments` branch passes its arguments to ArkWriter.write_next_utt in the wrong order
    (prepare_data.py:61: the ark path lands in `utt_id`, the segment name in `utt_mat`) and raises on the first
    segment; what it evidently means -- one utterance per segment, named by the segment -- is what happens here."""
    if not os.path.exists(featdir):
        os.makedirs(featdir)

    if os.path.isfile(datadir + '/segments'):
        segments = readfiles.read_segments(datadir + '/segments')
        found_segments = True
    else:
        print('''WARNING: no segments file found, assuming each wav file is
            seperate utterance''')
        found_segments = False

    # (the configuration is validated before the old feature file is removed: an unsupported transform length must not
    # cost the user the features that are already there)
    comp = feat.FeatureComputer(feat_type, dynamic, conf)
    if os.path.isfile(featdir + '/feats.ark'):
        os.remove(featdir + '/feats.ark')
    writer = ark.ArkWriter(featdir + '/feats.scp', featdir + '/feats.ark')
    wavfiles = readfiles.read_wavfiles(datadir + '/wav.scp')

    max_length = 0
    pending = {}  # sample rate -> [(utterance id, samples)]
    order = []    # utterance ids in the order the reference writes them
    done = {}

    def flush(rate=None):
        for r in ([rate] if rate is not None else list(pending)):
            items = pending.pop(r, [])
            if items:
                mats = comp.compute_batch([s for _, s in items], r, dtype=np.float32)
                for (uid, _), m in zip(items, mats):
                    done[uid] = m

    def drain():
        """write, in order, every leading utterance whose features are there"""
        nonlocal max_length
        while order and order[0] in done:
            uid = order.pop(0)
            m = done.pop(uid)
            writer.write_next_utt(uid, m)
            max_length = max(max_length, m.shape[0])

    def add(uid, rate, samples):
        order.append(uid)
        pending.setdefault(rate, []).append((uid, samples))
        if sum(s.size for _, s in pending[rate]) >= BATCH_SAMPLES:
            flush(rate)
            drain()

    for utt in wavfiles:
        rate, utterance = read_wav(wavfiles[utt])
        if found_segments:
            for seg in segments.get(utt, []):  # a recording without segments yields nothing
                add(seg[0], rate, utterance[int(seg[1] * rate):int(seg[2] * rate)])
        else:
            add(utt, rate, utterance)
    flush()
    drain()
    writer.close()

    copyfile(datadir + '/utt2spk', featdir + '/utt2spk')
    copyfile(datadir + '/spk2utt', featdir + '/spk2utt')
    copyfile(datadir + '/text', featdir + '/text')
    copyfile(datadir + '/wav.scp', featdir + '/wav.scp')

    with open(featdir + '/maxlength', 'w') as fid:
        fid.write(str(max_length))


def compute_cmvn(featdir):
    """per-speaker CMVN statistics -> cmvn.scp / cmvn.ark (prepare_data.py:80-118): one [2, dim+1] matrix per
    speaker of spk2utt, [[sum x | frame count], [sum x^2 | 0]]"""
    reader = ark.ArkReader(featdir + '/feats.scp')
    writer = ark.ArkWriter(featdir + '/cmvn.scp', featdir + '/cmvn.ark')
    names, batch, held = [], [], 0

    def flush():
        nonlocal held
        for name, stats in zip(names, device_features.cmvn_stats(batch)):
            writer.write_next_utt(name, stats)
        del names[:], batch[:]
        held = 0

    with open(featdir + '/spk2utt', 'r') as spk2utt:
        for line in spk2utt:
            split = line[0:len(line) - 1].split(' ')
            utts = [reader.read_utt(utt_id) for utt_id in split[1:len(split)]]
            names.append(split[0])
            batch.append(utts)
            held += sum(u.nbytes for u in utts)
            if held >= BATCH_CMVN_BYTES:
                flush()
    flush()
    writer.close()
    reader.close()


def shuffle_examples(featdir):
    """feats.scp in random order -> feats_shuffled.scp (prepare_data.py:120-139)"""
    with open(featdir + '/feats.scp', 'r') as featsfile:
        feats = featsfile.readlines()
    shuffle(feats)
    with open(featdir + '/feats_shuffled.scp', 'w') as feats_shuffledfile:
        feats_shuffledfile.writelines(feats)


def read_wav(wavfile):
    """(rate, samples) of a wav.scp entry (prepare_data.py:141-164): a file name, or a command line ending in `|`
    whose standard output is the wav file (the reference tees it through tmp.wav in the working directory; here it is
    read from the pipe)"""
    if wavfile[1]:
        cmd = wavfile[0].rstrip()
        if cmd.endswith('|'):
            cmd = cmd[:-1]
        data = subprocess.run(cmd, shell=True, check=True, stdout=subprocess.PIPE).stdout
        return wav.read(io.BytesIO(data))
    return wav.read(wavfile[0])
