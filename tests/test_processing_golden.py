"""processing/ (ark, feature_reader, target_coder, batchdispenser) against golden vectors produced by the
REFERENCE's own code (oracle/make_golden_io.py).  Bit-exact."""
import gzip
import io
import os
import shutil
from contextlib import redirect_stdout

import numpy as np
import pytest

from tfkaldi_amd.processing import ark, batchdispenser, feature_reader, readfiles, target_coder

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "io_golden.npz"))


@pytest.fixture()
def datadir(tmp_path):
    """materialise the reference-written files with real paths in the scp files"""
    d = str(tmp_path)
    for name in ("feats.ark", "cmvn.ark", "dbl.ark"):
        shutil.copy(os.path.join(GOLD, "io_%s.bin" % name), os.path.join(d, name))
    for name in ("feats.scp", "cmvn.scp"):
        text = open(os.path.join(GOLD, "io_%s.txt" % name)).read().replace("@DIR@", d)
        open(os.path.join(d, name), "w").write(text)
    shutil.copy(os.path.join(GOLD, "io_utt2spk.txt"), os.path.join(d, "utt2spk"))
    with gzip.open(os.path.join(d, "pdf.all.gz"), "wt") as f:
        f.write(open(os.path.join(GOLD, "io_pdf.all.txt")).read())
    return d


def same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


def test_splice_matches_reference(gold):
    assert same(feature_reader.splice(gold["ramp"], 1), gold["splice_ramp_c1"])
    for c in (0, 2, 5, 11):
        assert same(feature_reader.splice(gold["utt"], c), gold["splice_c%d" % c])
    # KAT 8c-5: zero-padded edges, and too-short utterances give None
    assert (gold["splice_ramp_c1"][0] == [0, 0, 0, 1, 2, 3]).all()
    assert (gold["splice_ramp_c1"][-1] == [8, 9, 10, 11, 0, 0]).all()
    assert feature_reader.splice(gold["ramp"][:2], 1) is None
    assert feature_reader.splice(gold["utt"], 12) is None  # 23 frames < 25


def test_cmvn_matches_reference(gold):
    assert same(feature_reader.apply_cmvn(gold["utt"], gold["cmvn_stats"]), gold["cmvn_out"])
    out = feature_reader.apply_cmvn(gold["ramp"], gold["cmvn_ramp_stats"])
    assert same(out, gold["cmvn_ramp_out"])
    assert np.allclose(np.abs(out[:, 0]), [1.46385, 0.87831, 0.29277, 0.29277, 0.87831, 1.46385], atol=1e-5)  # KAT 8c-6


def test_alignment_coder_matches_reference(gold):
    coder = target_coder.AlignmentCoder(lambda x, y: x, 50)
    enc = coder.encode(str(gold["ali_string"]))
    assert same(enc, gold["ali_encoded"]) and enc.dtype == np.uint32
    assert coder.num_labels == 50
    assert coder.decode(enc) == str(gold["ali_string"])
    with pytest.raises(KeyError):
        coder.encode("3 50")  # label outside the alphabet
    text = target_coder.TextCoder(lambda x, y: x)
    assert text.num_labels == 35 and (text.encode("<sos> a z <eos>") == [1, 9, 34, 0]).all()


def test_ark_writer_bytes_match_reference(gold, tmp_path):
    d = str(tmp_path)
    w = ark.ArkWriter(os.path.join(d, "feats.scp"), os.path.join(d, "feats.ark"))
    for i in range(6):
        w.write_next_utt("utt%02d" % i, gold["ark_utt%02d" % i])
    w.close()
    assert open(os.path.join(d, "feats.ark"), "rb").read() == open(os.path.join(GOLD, "io_feats.ark.bin"), "rb").read()
    assert open(os.path.join(d, "feats.scp")).read().replace(d, "@DIR@") == open(os.path.join(GOLD, "io_feats.scp.txt")).read()
    # KAT 8c-7: exact byte image of one 3x2 matrix
    w = ark.ArkWriter(os.path.join(d, "t.scp"), os.path.join(d, "t.ark"))
    w.write_next_utt("uttA", np.arange(6, dtype=np.float64).reshape(3, 2))
    w.close()
    raw = open(os.path.join(d, "t.ark"), "rb").read()
    assert raw[:19] == bytes.fromhex("75747441 00 42 46 4d 20 04 03000000 04 02000000".replace(" ", ""))
    assert len(raw) == 19 + 24 and open(os.path.join(d, "t.scp")).read() == "uttA %s:4\n" % os.path.join(d, "t.ark")


def test_ark_reader_matches_reference(gold, datadir):
    r = ark.ArkReader(os.path.join(datadir, "feats.scp"))
    seq = []
    for _ in range(8):
        uid, mat, looped = r.read_next_utt()
        assert same(mat, gold["ark_" + uid])
        seq.append("%s:%d" % (uid, int(looped)))
    assert seq == list(gold["reader_sequence"])
    assert same(r.read_utt("utt03"), gold["ark_utt03"])
    r = ark.ArkReader(os.path.join(datadir, "feats.scp"))
    r.read_next_utt(); r.read_next_utt()
    r.split()
    assert r.utt_ids == list(gold["reader_after_split"])  # drops the read ones AND the last one
    assert r.scp_position == 2                               # and leaves the cursor where it was
    # float64 archives
    open(os.path.join(datadir, "dbl.scp"), "w").write("dblutt %s:%d\n" % (os.path.join(datadir, "dbl.ark"), int(gold["ark_double_pos"])))
    _, dm, _ = ark.ArkReader(os.path.join(datadir, "dbl.scp")).read_next_utt()
    assert same(dm, gold["ark_double"]) and dm.dtype == np.float64
    # empty scp
    open(os.path.join(datadir, "empty.scp"), "w").close()
    assert ark.ArkReader(os.path.join(datadir, "empty.scp")).read_next_utt() == (None, None, True)


def test_ark_reader_rejects_compressed_and_text(datadir, capsys):
    p = os.path.join(datadir, "bad.ark")
    open(p, "wb").write(b"u\0BCM \x04\x01\x00\x00\x00\x04\x01\x00\x00\x00")
    open(os.path.join(datadir, "bad.scp"), "w").write("u %s:1\n" % p)
    with pytest.raises(SystemExit):
        ark.ArkReader(os.path.join(datadir, "bad.scp")).read_next_utt()
    assert "compressed" in capsys.readouterr().out
    open(p, "wb").write(b"u [\n 1 2 ]\n    ")
    with pytest.raises(SystemExit):
        ark.ArkReader(os.path.join(datadir, "bad.scp")).read_next_utt()
    assert "not binary" in capsys.readouterr().out


def test_batch_dispenser_matches_reference(gold, datadir):
    reader = feature_reader.FeatureReader(os.path.join(datadir, "feats.scp"), os.path.join(datadir, "cmvn.scp"),
                                          os.path.join(datadir, "utt2spk"), 2, 13)
    coder = target_coder.AlignmentCoder(lambda x, y: x, 20)
    disp = batchdispenser.AlignmentBatchDispenser(reader, coder, 2, os.path.join(datadir, "pdf.all.gz"))
    assert disp.num_utt == int(gold["disp_num_utt"]) == 5
    assert disp.num_batches == 2 == int(float(gold["disp_num_batches"]))  # Py2 floor of 5 / 2
    assert disp.max_target_length == int(gold["disp_max_target_length"])
    assert disp.max_input_length == 13 and disp.num_labels == 20
    buf = io.StringIO()
    with redirect_stdout(buf):
        for b in range(3):
            xs, ys = disp.get_batch()
            assert len(xs) == len(ys) == 2
            for j in range(2):
                assert same(xs[j], gold["disp_b%d_x%d" % (b, j)])
                assert same(ys[j], gold["disp_b%d_y%d" % (b, j)])
        disp.return_batch()
        assert same(disp.get_batch()[0][0], gold["disp_after_return_x0"])
        disp.skip_batch()
        assert same(disp.get_batch()[0][0], gold["disp_after_skip_x0"])
        assert (disp.compute_target_count() == gold["disp_target_count"]).all()
    assert buf.getvalue() == str(gold["disp_warnings"])


def test_read_utt2spk(gold, datadir):
    got = readfiles.read_utt2spk(os.path.join(datadir, "utt2spk"))
    assert sorted(got.items()) == [tuple(x) for x in gold["utt2spk_keys"]]


def test_text_targets_match_reference():
    """aurora4_normalizer + TextCoder against tests/golden/text_golden.json (oracle/make_golden_text.py: the
    reference's own code run on these transcriptions)"""
    import json
    from tfkaldi_amd.processing import target_normalizers
    with open(os.path.join(GOLD, "text_golden.json")) as fid:
        gold = json.load(fid)
    coder = target_coder.TextCoder(target_normalizers.aurora4_normalizer)
    assert coder.alphabet == gold["alphabet"] and coder.num_labels == gold["num_labels"] == 35
    for case in gold["cases"]:
        assert target_normalizers.aurora4_normalizer(case["text"], coder.lookup.keys()) == case["normalized"]
        got = coder.encode(case["text"])
        assert got.dtype == np.uint32 and got.tolist() == case["encoded"]
    assert coder.decode(coder.encode("AB C")) == "<sos> a b <space> c <eos>"
