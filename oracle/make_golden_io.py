"""Generate tests/golden/io_*.npz|.bin from the REFERENCE's own numpy code (run in the build container only).

The reference's processing/ package is Python 2; it is converted with lib2to3 into a scratch directory under
/tmp (never into this repository) with mechanical shims -- numpy-2 rejects
`np.set_printoptions(threshold=np.nan)` (processing/ark.py:25-26), struct/bytes comparisons need
bytes literals under Python 3 (ark.py:73-76, 204-206), and Python 2's flooring int `/` in base.py becomes `//`.  The fixtures are DATA: inputs and the outputs the
reference produced for them.

    python oracle/make_golden_io.py          # needs /root/reference; writes tests/golden/
"""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "tests", "golden")


def import_reference_processing():
    scratch = tempfile.mkdtemp(prefix="tfkaldi_ref_", dir="/tmp")
    dst = os.path.join(scratch, "processing")
    shutil.copytree(os.path.join(REF, "processing"), dst, ignore=shutil.ignore_patterns("__pycache__"))
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", dst], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    ark = os.path.join(dst, "ark.py")
    s = open(ark).read()
    s = s.replace("np.set_printoptions(threshold=np.nan)", "").replace("np.set_printoptions(linewidth=np.nan)", "")
    s = s.replace('header[0] != "B"', 'header[0] != b"B"').replace('header[1] == "C"', 'header[1] == b"C"')
    s = s.replace('header[1] == "F"', 'header[1] == b"F"').replace('header[1] == "D"', 'header[1] == b"D"')
    s = s.replace("struct.pack('<%ds'%(len(utt_id)), utt_id)", "struct.pack('<%ds'%(len(utt_id)), utt_id.encode())")
    s = s.replace("struct.pack('<xcccc', 'B', 'F', 'M', ' ')", "struct.pack('<xcccc', b'B', b'F', b'M', b' ')")
    open(ark, "w").write(s)
    bd = os.path.join(dst, "batchdispenser.py")
    s = open(bd).read()
    s = s.replace("with gzip.open(target_path, 'rb') as fid:", "with gzip.open(target_path, 'rt') as fid:")
    open(bd, "w").write(s)
    # base.py relies on Python 2's flooring `/` between ints (`nfft/2+1`, `samplerate/2`: base.py:76,134,151,205-217)
    bp = os.path.join(dst, "base.py")
    s = open(bp).read()
    s = s.replace("nfft/2+1", "nfft//2+1").replace("samplerate/2", "samplerate//2")
    open(bp, "w").write(s)
    sys.path.insert(0, scratch)
    sys.path.insert(0, dst)  # the package uses implicit relative imports (import ark, import readfiles)
    import processing.ark as ark_mod  # noqa
    import processing.batchdispenser as bd_mod  # noqa
    import processing.feature_reader as fr_mod  # noqa
    import processing.target_coder as tc_mod  # noqa
    import processing.readfiles as rf_mod  # noqa
    return scratch, ark_mod, fr_mod, bd_mod, tc_mod, rf_mod


def main():
    os.makedirs(GOLD, exist_ok=True)
    scratch, ark, fr, bd, tc, rf = import_reference_processing()
    rng = np.random.default_rng(2024)
    out = {}

    # ---- splice (feature_reader.py:117-156) and apply_cmvn (:91-115) ----
    ramp = np.arange(12, dtype=np.float32).reshape(6, 2)
    out["splice_ramp_c1"] = fr.splice(ramp, 1)
    out["ramp"] = ramp
    utt = rng.standard_normal((23, 5)).astype(np.float32) * 2 + 5
    out["utt"] = utt
    for c in (0, 2, 5, 11):
        out["splice_c%d" % c] = fr.splice(utt, c)
    assert fr.splice(ramp[:2], 1) is None  # shorter than 2c+1 frames -> None (feature_reader.py:132-133)
    stats = np.zeros((2, 6), dtype=np.float32)
    stats[0, :5] = utt.sum(0); stats[0, 5] = utt.shape[0]; stats[1, :5] = (utt ** 2).sum(0)
    out["cmvn_stats"] = stats
    out["cmvn_out"] = fr.apply_cmvn(utt, stats)
    rstats = np.zeros((2, 3), dtype=np.float32)
    rstats[0, :2] = ramp.sum(0); rstats[0, 2] = 6; rstats[1, :2] = (ramp ** 2).sum(0)
    out["cmvn_ramp_stats"] = rstats
    out["cmvn_ramp_out"] = fr.apply_cmvn(ramp, rstats)

    # ---- AlignmentCoder.encode (target_coder.py:36-55, 120-142) ----
    coder = tc.AlignmentCoder(lambda x, y: x, 50)
    ali = "3 3 17 49 0 0 12"
    out["ali_encoded"] = coder.encode(ali)
    out["ali_string"] = np.array(ali)

    # ---- ArkWriter / ArkReader bytes (ark.py:37-211) ----
    d = os.path.join(scratch, "io")
    os.makedirs(d)
    w = ark.ArkWriter(os.path.join(d, "feats.scp"), os.path.join(d, "feats.ark"))
    mats = {}
    order = []
    for i, n in enumerate((7, 13, 3, 9, 11, 8)):
        uid = "utt%02d" % i
        m = rng.standard_normal((n, 4)).astype(np.float32)
        mats[uid] = m
        order.append(uid)
        w.write_next_utt(uid, m)
    w.close()
    ark_bytes = open(os.path.join(d, "feats.ark"), "rb").read()
    scp_text = open(os.path.join(d, "feats.scp")).read().replace(d, "@DIR@")
    open(os.path.join(GOLD, "io_feats.ark.bin"), "wb").write(ark_bytes)
    open(os.path.join(GOLD, "io_feats.scp.txt"), "w").write(scp_text)
    for uid in order:
        out["ark_" + uid] = mats[uid]
    # a float64 ('DM') matrix, which the reference reader accepts (ark.py:86-90)
    import struct
    with open(os.path.join(d, "dbl.ark"), "wb") as f:
        f.write(b"dblutt")
        pos = f.tell()
        f.write(struct.pack("<xcccc", b"B", b"D", b"M", b" "))
        f.write(struct.pack("<bi", 4, 3)); f.write(struct.pack("<bi", 4, 2))
        dm = np.arange(6, dtype=np.float64).reshape(3, 2) / 7
        f.write(dm.tobytes())
    open(os.path.join(d, "dbl.scp"), "w").write("dblutt %s:%d\n" % (os.path.join(d, "dbl.ark"), pos))
    r = ark.ArkReader(os.path.join(d, "dbl.scp"))
    _, got, _ = r.read_next_utt()
    out["ark_double"] = got
    open(os.path.join(GOLD, "io_dbl.ark.bin"), "wb").write(open(os.path.join(d, "dbl.ark"), "rb").read())
    out["ark_double_pos"] = np.array(pos)

    # reader sequencing: wrap-around flag, split() (drops what was read AND the last entry: ark.py:161-165)
    r = ark.ArkReader(os.path.join(d, "feats.scp"))
    seq = []
    for _ in range(8):
        uid, m, looped = r.read_next_utt()
        seq.append("%s:%d" % (uid, int(looped)))
    out["reader_sequence"] = np.array(seq)
    r = ark.ArkReader(os.path.join(d, "feats.scp"))
    r.read_next_utt(); r.read_next_utt()
    r.split()
    out["reader_after_split"] = np.array(r.utt_ids)

    # ---- FeatureReader + AlignmentBatchDispenser sequencing (batchdispenser.py:31-161, 200-223) ----
    spk = {uid: "spk%d" % (i % 2) for i, uid in enumerate(order)}
    open(os.path.join(d, "utt2spk"), "w").write("".join("%s %s\n" % (u, spk[u]) for u in order))
    cw = ark.ArkWriter(os.path.join(d, "cmvn.scp"), os.path.join(d, "cmvn.ark"))
    for s in ("spk0", "spk1"):
        rows = np.concatenate([mats[u] for u in order if spk[u] == s])
        st = np.zeros((2, 5), dtype=np.float32)
        st[0, :4] = rows.sum(0); st[0, 4] = rows.shape[0]; st[1, :4] = (rows ** 2).sum(0)
        cw.write_next_utt(s, st)
    cw.close()
    open(os.path.join(GOLD, "io_cmvn.ark.bin"), "wb").write(open(os.path.join(d, "cmvn.ark"), "rb").read())
    open(os.path.join(GOLD, "io_cmvn.scp.txt"), "w").write(open(os.path.join(d, "cmvn.scp")).read().replace(d, "@DIR@"))
    open(os.path.join(GOLD, "io_utt2spk.txt"), "w").write(open(os.path.join(d, "utt2spk")).read())
    # alignments: utt02 (3 frames) is too short for context 2; utt04 has no alignment
    ali_lines = []
    for u in order:
        if u == "utt04":
            continue
        ali_lines.append("%s %s" % (u, " ".join(str(x) for x in rng.integers(0, 20, size=mats[u].shape[0]))))
    with gzip.open(os.path.join(d, "pdf.all.gz"), "wt") as f:
        f.write("\n".join(ali_lines) + "\n")
    open(os.path.join(GOLD, "io_pdf.all.txt"), "w").write("\n".join(ali_lines) + "\n")
    reader = fr.FeatureReader(os.path.join(d, "feats.scp"), os.path.join(d, "cmvn.scp"), os.path.join(d, "utt2spk"), 2, 13)
    coder = tc.AlignmentCoder(lambda x, y: x, 20)
    disp = bd.AlignmentBatchDispenser(reader, coder, 2, os.path.join(d, "pdf.all.gz"))
    out["disp_num_batches"] = np.array(disp.num_batches)  # Python-3 true division here; Py2 floors (len 5 / 2 = 2)
    out["disp_num_utt"] = np.array(disp.num_utt)
    out["disp_max_target_length"] = np.array(disp.max_target_length)
    import io
    import contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        for b in range(3):
            xs, ys = disp.get_batch()
            for j, (x, y) in enumerate(zip(xs, ys)):
                out["disp_b%d_x%d" % (b, j)] = x
                out["disp_b%d_y%d" % (b, j)] = y
        disp.return_batch()
        xs, ys = disp.get_batch()
        out["disp_after_return_x0"] = xs[0]
        disp.skip_batch()
        xs, ys = disp.get_batch()
        out["disp_after_skip_x0"] = xs[0]
        out["disp_target_count"] = disp.compute_target_count()
    out["disp_warnings"] = np.array(buf.getvalue())
    out["utt2spk_keys"] = np.array(sorted(rf.read_utt2spk(os.path.join(d, "utt2spk")).items()))

    np.savez(os.path.join(GOLD, "io_golden.npz"), **out)
    shutil.rmtree(scratch)
    print("wrote %d arrays to %s" % (len(out), os.path.abspath(GOLD)))


if __name__ == "__main__":
    main()
