"""The packed feed (BatchDispenser.next_packed / prefetch, Trainer.update_packed): the same batches in the same order
as the reference's get_batch, a data-parallel rank touching only its own utterances' bytes, buffers that stay valid
for as long as a caller holds them.  CPU only; the GPU side is tests/test_gpu_packed_feed.py."""
import io
import os
import socket
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest

from tfkaldi_amd import synthetic
from tfkaldi_amd.dataparallel import partition
from tfkaldi_amd.neuralNetworks.trainer import MicrobatchSelector, Trainer, microbatch_indices
from tfkaldi_amd.processing import batchdispenser, feature_reader, target_coder
from tfkaldi_amd.processing.batchdispenser import select_all
from tfkaldi_amd.processing.feature_reader import Unspliced

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTEXT, DIM, PDFS = 2, 6, 11


def _corpus(tmp_path, num_utt=41, seed=5):
    rng = np.random.default_rng(seed)
    lengths = rng.integers(3, 19, size=num_utt)  # some shorter than 2 * CONTEXT + 1 = 5: skipped with a WARNING
    lengths[7] = 2
    paths = synthetic.write_corpus(str(tmp_path), num_utt, PDFS, feat_dim=DIM, lengths=lengths, num_speakers=3)
    return paths, lengths


def _dispenser(paths, size, drop=()):
    reader = feature_reader.FeatureReader(paths["feats_scp"], paths["cmvn_scp"], paths["utt2spk"], CONTEXT, 18)
    disp = batchdispenser.AlignmentBatchDispenser(reader, target_coder.AlignmentCoder(lambda x, y: x, PDFS), size,
                                                  paths["alignments"])
    for utt in drop:  # utterances without targets: the other WARNING of get_batch
        del disp.target_dict[utt]
    return disp


def _unpack(batch):
    """the (inputs, targets) lists get_batch would have returned for the utterances a PackedBatch holds"""
    xs, ys, row, tgt = [], [], 0, 0
    for u, (n, m) in enumerate(zip(batch.lens, batch.target_lens)):
        cmvn = None if batch.cmvn is None else batch.cmvn[u]
        xs.append(Unspliced(batch.frames[row:row + n], batch.context_width, cmvn=cmvn).spliced())
        ys.append(batch.targets[tgt:tgt + m])
        row, tgt = row + n, tgt + m
    return xs, ys


def _same(a, b):
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def test_packed_batches_are_the_reference_batches(tmp_path):
    """next_packed(select_all) == get_batch: frames (after the deferred CMVN + splice, bit for bit), targets, WARNING
    lines, wrap-around at the end of the scp -- over more than one epoch"""
    paths, _ = _corpus(tmp_path)
    drop = ("utt000004", "utt000020")
    a, b = _dispenser(paths, 6, drop), _dispenser(paths, 6, drop)
    out_a, out_b = io.StringIO(), io.StringIO()
    for step in range(17):
        with redirect_stdout(out_a):
            xs, ys = a.get_batch()
        with redirect_stdout(out_b):
            batch = b.next_packed(select_all)
        assert batch.batch_utts == 6 and batch.groups == [(0, 6, 0, batch.num_frames, 0, batch.targets.size)]
        assert batch.frames.dtype == np.float32 and batch.frames.flags.c_contiguous
        assert batch.targets.dtype == np.int32 and batch.lens.dtype == np.int32
        xs2, ys2 = _unpack(batch)
        assert len(xs2) == 6
        for x, x2, y, y2 in zip(xs, xs2, ys, ys2):
            assert _same(x, x2) and (y == y2).all() and y.dtype == np.uint32
    assert out_a.getvalue() == out_b.getvalue()
    assert "WARNING no targets for utt000004" in out_a.getvalue()
    assert "WARNING utt000007 is too short to splice" in out_a.getvalue()


def test_prefetch_never_changes_the_sequence(tmp_path):
    """prefetch + return_batch / skip_batch / split / get_batch in between: every batch handed out is the one a
    dispenser without prefetch hands out at the same point, and the WARNING lines appear at hand-out time"""
    paths, _ = _corpus(tmp_path)
    a, b = _dispenser(paths, 5), _dispenser(paths, 5)
    script = ["get", "get", "split", "get", "get", "return", "get", "skip", "get", "return", "return", "get", "get",
              "mix", "get", "skip", "skip", "get"]
    out_a, out_b = io.StringIO(), io.StringIO()
    for op in script:
        if op in ("get", "mix"):
            with redirect_stdout(out_a):
                xs, ys = a.get_batch()
            with redirect_stdout(out_b):
                if op == "mix":
                    xs2, ys2 = b.get_batch()  # the reference call while a prefetched batch is waiting
                else:
                    batch = b.next_packed(select_all)
                    xs2, ys2 = _unpack(batch)
            assert all(_same(x, x2) and (y == y2).all() for x, x2, y, y2 in zip(xs, xs2, ys, ys2))
            assert out_a.getvalue() == out_b.getvalue()  # (nothing printed early by the prefetch below)
            with redirect_stdout(out_b):
                b.prefetch(select_all)
                b.prefetch(select_all)  # a second call is a no-op
            assert out_a.getvalue() == out_b.getvalue()
        else:
            for d in (a, b):
                {"split": d.split, "return": d.return_batch, "skip": d.skip_batch}[op]()
        assert a.feature_reader.reader.utt_ids == b.feature_reader.reader.utt_ids
    assert "too short to splice" in out_a.getvalue()


def test_a_held_batch_stays_valid_and_buffers_are_recycled(tmp_path):
    paths, _ = _corpus(tmp_path)
    disp = _dispenser(paths, 6)
    held = disp.next_packed(select_all)
    snapshot = held.frames.copy()
    views = [held.frames[:3], held.frames[3:]]
    seen = set()
    for _ in range(12):
        batch = disp.next_packed(select_all)
        seen.add(batch.frames.__array_interface__["data"][0])
        del batch
    assert (held.frames == snapshot).all()
    assert len(seen) <= 2  # released buffers come back from the pool instead of being allocated again
    base = held.frames.__array_interface__["data"][0]
    del held
    assert (np.concatenate(views) == snapshot).all()  # ... but only once the LAST view has gone
    assert base not in {disp.next_packed(select_all).frames.__array_interface__["data"][0] for _ in range(3)}
    del views
    assert base in {disp.next_packed(select_all).frames.__array_interface__["data"][0] for _ in range(3)}


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("size,per_minibatch", [(16, 2), (13, 4), (5, 8)])
def test_a_rank_reads_only_its_own_utterances(tmp_path, world, size, per_minibatch):
    """the selector deals the reference's micro-batches (the len % U quirk included) to the ranks in contiguous blocks;
    every rank's ArkReader fetches exactly the bytes of its own utterances; together the ranks cover the step"""
    paths, lengths = _corpus(tmp_path, num_utt=60)
    full = _dispenser(paths, size)
    ranks = [_dispenser(paths, size) for _ in range(world)]
    selectors = [MicrobatchSelector(per_minibatch, r, world) for r in range(world)]
    stub = type("T", (), {"loss_kind": "cross_enthropy"})()
    for step in range(4):
        xs, ys = full.get_batch()
        plan = [idx for idx in microbatch_indices(size, per_minibatch) if idx]
        covered = []
        for r, (disp, sel) in enumerate(zip(ranks, selectors)):
            before = disp.feature_reader.reader.bytes_read
            batch = disp.next_packed(sel)
            total, start, end = batch.info
            assert total == len(plan) and (start, end) == partition(len(plan), world)[r]
            mine = plan[start:end]
            own = [i for idx in mine for i in idx]
            assert disp.feature_reader.reader.bytes_read - before == sum(xs[i].shape[0] for i in own) * DIM * 4
            assert len(batch.groups) == len(mine) and batch.batch_utts == size
            mbs = Trainer._packed_microbatches(stub, batch)
            for mb, idx in zip(mbs, mine):
                want_x = np.concatenate([xs[i] for i in idx])
                got_x = np.concatenate([Unspliced(mb.raw[a:b], mb.context_width, cmvn=mb.cmvn[u]).spliced()
                                        for u, (a, b) in enumerate(zip(np.cumsum(mb.lens) - mb.lens, np.cumsum(mb.lens)))])
                assert _same(want_x, got_x)
                assert (mb.y == np.concatenate([ys[i] for i in idx])).all()
            covered += mine
            # every rank's cursor is where the full reader's is: the next batch starts at the same utterance
            assert disp.feature_reader.reader.scp_position == full.feature_reader.reader.scp_position
        assert covered == plan


def test_float64_archives_fall_back_to_host_normalisation(tmp_path):
    """statistics stored as float64 (an archive Kaldi wrote): the host normalises in float64 as the reference does and
    the packed batch carries normalised frames with identity (or no) CMVN rows -- same values as get_batch"""
    from tfkaldi_amd.processing import ark
    paths, _ = _corpus(tmp_path)
    src = ark.ArkReader(paths["cmvn_scp"])
    d = str(tmp_path / "dbl")
    os.makedirs(d)
    with open(os.path.join(d, "cmvn.ark"), "wb") as fh, open(os.path.join(d, "cmvn.scp"), "w") as scp:
        for spk in src.utt_ids:
            mat = src.read_utt(spk).astype(np.float64)
            fh.write(spk.encode())
            scp.write("%s %s:%d\n" % (spk, os.path.join(d, "cmvn.ark"), fh.tell()))
            fh.write(b"\0BDM " + b"\x04" + np.int32(mat.shape[0]).tobytes() + b"\x04" + np.int32(mat.shape[1]).tobytes())
            fh.write(mat.tobytes())
    paths = dict(paths, cmvn_scp=os.path.join(d, "cmvn.scp"))
    a, b = _dispenser(paths, 4), _dispenser(paths, 4)
    for _ in range(3):
        with redirect_stdout(io.StringIO()):
            xs, ys = a.get_batch()
            batch = b.next_packed(select_all)
        assert batch.cmvn is None
        xs2, _ = _unpack(batch)
        assert all(_same(x, x2) for x, x2 in zip(xs, xs2))


# ---- two gloo ranks, each with its own dispenser over the same files, against the serial reference loop ----
KW = dict(input_dim=DIM * (2 * CONTEXT + 1), num_layers=2, num_units=10, output_dim=PDFS, nonlin="relu", batch_norm=True,
          init_learning_rate=1e-2, num_steps=10)


def _oracle():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.dnn_oracle import OracleDNN
    from util import randomize
    o = OracleDNN(**KW)
    randomize(o, np.random.default_rng(42))
    return o


class _RawOracleEngine(object):
    """tests/oracle_engine.OracleEngine + the raw entry point: CMVN and splice through the host functions"""

    def __new__(cls, oracle):
        from oracle_engine import OracleEngine

        class Raw(OracleEngine):
            def _spliced(self, raw, lens, context_width, cmvn):
                rows = np.cumsum(lens) - lens
                return np.concatenate([Unspliced(raw[a:a + n], context_width, cmvn=None if cmvn is None else cmvn[u]).spliced()
                                       for u, (a, n) in enumerate(zip(rows, lens))])

            def accumulate_raw(self, raw, y, lens, context_width, last=False, cmvn=None):
                self.accumulate(self._spliced(raw, lens, context_width, cmvn), y, last=last)

            def eval_accumulate_raw(self, raw, y, lens, context_width, cmvn=None):
                self.eval_accumulate(self._spliced(raw, lens, context_width, cmvn), y)

        return Raw(oracle)


def _worker(rank, world, port, corpus_dir, out_dir, size, per_minibatch, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from tfkaldi_amd.dataparallel import DataParallel, init_from_env
    init_from_env()
    dp = DataParallel(mode="allreduce")
    eng = _RawOracleEngine(_oracle())
    paths = {k: os.path.join(corpus_dir, v) for k, v in dict(
        feats_scp="feats.scp", cmvn_scp="cmvn.scp", utt2spk="utt2spk", alignments="pdf.all").items()}
    disp = _dispenser(paths, size)
    select = MicrobatchSelector(per_minibatch, rank, world)
    stub = type("T", (), {"loss_kind": "cross_enthropy"})()
    with redirect_stdout(io.StringIO()):
        valid = disp.next_packed(select, 2 * size)
        disp.split()
        losses = []
        for step in range(steps):
            batch = disp.next_packed(select)
            total, _, end = batch.info
            losses.append(dp.train_own(eng, Trainer._packed_microbatches(stub, batch), total - end,
                                       overlap=lambda: disp.prefetch(select)))
            if step == 1:  # a rollback by one batch with a prefetched batch waiting
                disp.return_batch()
        losses.append(dp.eval_own(eng, Trainer._packed_microbatches(stub, valid)))
    eng.sync_params()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), losses=np.array(losses),
             bytes_read=disp.feature_reader.reader.bytes_read, **eng.o.params())
    dist.destroy_process_group()


@pytest.mark.parametrize("size,per_minibatch", [(8, 2), (7, 2)])
def test_two_ranks_with_rank_local_reads_equal_the_serial_loop(tmp_path, size, per_minibatch):
    import torch.multiprocessing as mp
    corpus = tmp_path / "corpus"
    paths, _ = _corpus(corpus, num_utt=50)
    out = tmp_path / "out"
    out.mkdir()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    steps, world = 4, 2
    mp.spawn(_worker, args=(world, port, str(corpus), str(out), size, per_minibatch, steps), nprocs=world, join=True)
    # the serial loop of the reference: whole batches through get_batch and the trainer's host micro-batches
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    serial = _oracle()
    disp = _dispenser(paths, size)
    with redirect_stdout(io.StringIO()):
        held = [disp.get_batch() for _ in range(2)]
        disp.split()
        vx = [x for xs, _ in held for x in xs]
        vy = [y for _, ys in held for y in ys]
        want, total_bytes = [], 0
        for step in range(steps):
            xs, ys = disp.get_batch()
            for idx in microbatch_indices(len(xs), per_minibatch):
                if idx:
                    serial.accumulate(np.concatenate([xs[i] for i in idx]), np.concatenate([ys[i] for i in idx]).astype(np.int32))
                    total_bytes += sum(xs[i].shape[0] for i in idx) * DIM * 4
            want.append(serial.apply())
            if step == 1:
                disp.return_batch()
        for idx in microbatch_indices(len(vx), per_minibatch):
            if idx:
                serial.eval_accumulate(np.concatenate([vx[i] for i in idx]), np.concatenate([vy[i] for i in idx]).astype(np.int32))
                total_bytes += sum(vx[i].shape[0] for i in idx) * DIM * 4
        want.append(serial.eval_finish())
    got = [np.load(str(out / ("rank%d.npz" % r))) for r in range(world)]
    for g in got:
        assert np.allclose(g["losses"], want, rtol=1e-12, atol=0)
        for k, v in serial.params().items():
            assert np.allclose(g[k], v, rtol=1e-9, atol=1e-12), k
    # together the ranks fetched every used utterance exactly once (+ the one batch each prefetched past the end, and
    # the prefetched batch the rollback put back)
    fetched = sum(int(g["bytes_read"]) for g in got)
    assert total_bytes <= fetched <= total_bytes + 3 * size * 18 * DIM * 4
    assert max(int(g["bytes_read"]) for g in got) < 0.75 * fetched
