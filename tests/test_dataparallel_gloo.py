"""The N > 1 path on CPU: two processes, gloo backend, the product's DataParallel class driving oracle-backed
engines.  Checks KAT 8c-3: k serial micro-batches == the same micro-batches sharded over ranks + SUM
all-reduce, including uneven shards, an idle rank, BN moving averages and the validation loss."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(input_dim=12, num_layers=2, num_units=10, output_dim=6, nonlin="relu", batch_norm=True,
          init_learning_rate=1e-2, num_steps=10)


def _data(num_mb, seed=0):
    rng = np.random.default_rng(seed)
    return [((rng.standard_normal((9 + 3 * i, KW["input_dim"]))).astype(np.float32),
             rng.integers(0, KW["output_dim"], size=9 + 3 * i).astype(np.int32)) for i in range(num_mb)]


def _oracle():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.dnn_oracle import OracleDNN
    from util import randomize
    o = OracleDNN(**KW)
    randomize(o, np.random.default_rng(42))
    return o


def _worker(rank, world, port, num_mb, out_dir, mode="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_engine import OracleEngine
    from tfkaldi_amd.dataparallel import DataParallel, init_from_env
    assert init_from_env() == (rank, world, 0)
    import tfkaldi_amd.dataparallel as dpmod
    dpmod.BucketReducer.MIN_SHARD_FLOATS = 8  # the test net is tiny: shard every span that divides
    dp = DataParallel(mode=mode)
    assert dp.enabled and dp.world == world and dp.rank == rank
    eng = OracleEngine(_oracle())
    losses = []
    for step in range(2):
        losses.append(dp.train_step(eng, _data(num_mb, seed=step)))
        if mode == "sharded" and num_mb >= world:
            assert "rs" in dp.last_kinds, dp.last_kinds  # the sharded exchange really ran
    losses.append(dp.eval_step(eng, _data(num_mb, seed=7)))
    eng.sync_params()
    o = eng.o
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), losses=np.array(losses),
             mov_mean=np.stack(o.mov_mean), mov_var=np.stack(o.mov_var), **o.params())
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("mode", ["sharded", "allreduce"])
@pytest.mark.parametrize("num_mb", [4, 3, 1])  # even shards, uneven shards, one idle rank
def test_two_ranks_equal_serial(tmp_path, num_mb, mode):
    """both exchange steps of BucketReducer: reduce-scatter -> Adam on the rank's share -> all-gather of the
    parameters ("sharded") and all-reduce -> full Adam on every rank ("allreduce")"""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), num_mb, str(tmp_path), mode), nprocs=world, join=True)
    serial = _oracle()
    want = []
    for step in range(2):
        for X, y in _data(num_mb, seed=step):
            serial.accumulate(X, y)
        want.append(serial.apply())
    for X, y in _data(num_mb, seed=7):
        serial.eval_accumulate(X, y)
    want.append(serial.eval_finish())
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert np.allclose(got["losses"], want, rtol=1e-12, atol=0)
        for k, v in serial.params().items():
            assert np.allclose(got[k], v, rtol=1e-9, atol=1e-12), k
        assert np.allclose(got["mov_mean"], np.stack(serial.mov_mean), rtol=1e-10, atol=1e-14)
        assert np.allclose(got["mov_var"], np.stack(serial.mov_var), rtol=1e-10, atol=1e-14)


def _fallback_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tfkaldi_amd.dataparallel import DataParallel, init_from_env
    import tfkaldi_amd.dataparallel as dpmod
    init_from_env()
    closed = []

    entered = []

    class Flaky(object):  # stands in for NativeExchange: comes up on rank 0 only
        native = True

        def __init__(self, engine, group, mode=None):
            entered.append(True)
            if dist.get_rank() == 1:
                raise RuntimeError("ncclCommInitRank failed: unhandled system error")
            engine.on_close.insert(0, self.close)
            engine.param_access_hook = lambda: None

        def close(self):
            closed.append(True)

    class Lib(object):  # the one library call the agreement makes before anything collective
        def __init__(self, loadable):
            self.loadable = loadable

        def tfk_comm_available(self, handle, mode):
            if not self.loadable:
                raise RuntimeError("RCCL could not be loaded: no such file")
            return 0

    class Cfg(object):
        device = 0

    class Eng(object):
        def __init__(self, loadable=True):
            self.on_close, self.param_access_hook = [], None
            self.lib, self._h, self.cfg = Lib(loadable), None, Cfg()

    dpmod.NativeExchange = Flaky
    # (a) rank 1 cannot even load RCCL: the ranks hear of it BEFORE anyone enters the collective bootstrap (the real
    # NativeExchange would block in ncclCommInitRank waiting for rank 1) -- nobody constructs one
    dp = DataParallel(mode="sharded")
    eng = Eng(loadable=rank != 1)
    got = dp._native_exchange(eng)
    assert got is None and dp.native_failure and entered == [] and closed == []
    assert ("could not be loaded" in dp.native_failure) == (rank == 1)
    # (b) every rank can load it, rank 1's communicator is refused: every rank agrees, nobody keeps the in-library exchange,
    # the rank that had one gives it back, everyone knows why
    dp = DataParallel(mode="sharded")
    eng = Eng()
    got = dp._native_exchange(eng)
    assert got is None and dp.native_failure and entered == [True]
    assert eng.on_close == [] and eng.param_access_hook is None
    assert closed == ([True] if rank == 0 else [])
    assert ("ncclCommInitRank failed" in dp.native_failure) == (rank == 1)
    os.environ["TFK_DP_COMM"] = "native-only"
    try:
        dp._native_exchange(Eng())
    except RuntimeError as e:
        assert "in-library exchange unavailable" in str(e)
    else:
        raise AssertionError("native-only must make the failure fatal")
    open(os.path.join(out_dir, "ok%d" % rank), "w").close()
    dist.destroy_process_group()


def test_ranks_agree_to_leave_the_in_library_exchange(tmp_path):
    """one rank cannot bring the in-library RCCL exchange up: ALL ranks run the torch.distributed reducer.  Two agreements
    (flags all-reduced with MIN): a local probe BEFORE the collective bootstrap -- a rank that cannot load RCCL must not leave
    the others blocked inside ncclCommInitRank -- and the outcome of the creation itself, after which the rank that did create
    one closes it; TFK_DP_COMM=native-only turns the agreement into an error"""
    mp.spawn(_fallback_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok0")) and os.path.exists(os.path.join(str(tmp_path), "ok1"))


def _overlap_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_engine import OracleEngine
    from tfkaldi_amd.dataparallel import DataParallel, init_from_env, partition
    init_from_env()
    dp = DataParallel(mode="allreduce")
    eng = OracleEngine(_oracle())
    order = []

    def overlap():
        order.append("overlap")
        if rank == 1:
            raise IOError("short read of the next batch")

    red = dp.reducer(eng)
    orig = red.finish_and_apply

    def spy(engine, ov=None):
        # the hook runs INSIDE finish_and_apply -- behind everything the step launches, in front of the wait -- not before it
        order.append("finish_and_apply")
        return orig(engine, ov)
    red.finish_and_apply = spy
    mbs = _data(4, seed=0)
    start, end = partition(len(mbs), world)[rank]
    try:
        loss = dp.train_own(eng, mbs[start:end], len(mbs) - end, overlap)
        raised = None
    except IOError as exc:
        loss, raised = None, str(exc)
    assert order == ["finish_and_apply", "overlap"], order
    assert (raised is not None) == (rank == 1), raised
    # the step itself completed on BOTH ranks (the failing rank took part in every collective before it re-raised): the next
    # one runs, and the replicas still agree
    loss2 = dp.train_step(eng, _data(4, seed=1))
    eng.sync_params()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), loss2=loss2, **eng.o.params())
    dist.destroy_process_group()


def test_a_failing_overlap_hook_does_not_strand_the_peers(tmp_path):
    """round-4 advisor: `overlap()` -- the dispenser's prefetch -- ran before the tail collectives and the optimiser were launched,
    and an exception in it (a short read) unwound one rank out of the step while its peers waited in their collectives.  Now it
    runs once the whole step is enqueued, and what it raises is re-raised after the step has completed on every rank."""
    world = 2
    mp.spawn(_overlap_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = (np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world))
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
