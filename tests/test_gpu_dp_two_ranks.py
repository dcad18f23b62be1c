"""The product's data-parallel path with REAL HIP engines: 2, 4 or 8 processes (sharing the one GPU of the test box,
gloo transport because RCCL refuses two ranks per device) run DataParallel.train_step / eval_step -- engine
bucket callback -> async collectives on views of the engine's reduce region -> optimiser per span -> parameter
gathers consumed layer by layer by the next forward pass -- and must reproduce the single-process result that
processes all micro-batches serially (KAT 8c-3).  gloo cannot reduce-scatter device tensors: with TFK_DP_EMULATE_RS=1
the reducer emulates that one collective by an all-reduce and runs the rest of the sharded protocol (Adam on the
rank's 1/world of every span, all-gather of the parameters or of the bf16 shadow, sharded fp32 masters, replica
checksum) exactly as it does over RCCL.  One case runs eight ranks at BASELINE cfg2's size (26 M parameters)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(input_dim=40, num_layers=2, num_units=64, output_dim=24, nonlin="relu", batch_norm=True,
          init_learning_rate=1e-3, num_steps=10)


def _data(num_mb, seed):
    rng = np.random.default_rng(seed)
    return [((rng.standard_normal((50 + 9 * i, KW["input_dim"]))).astype(np.float32),
             rng.integers(0, KW["output_dim"], size=50 + 9 * i).astype(np.int32)) for i in range(num_mb)]


def _engine(torch_state, dtype="float32", kw=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import make_pair
    eng, _ = make_pair(np.random.default_rng(3), torch_state=torch_state, compute_dtype=dtype, **(kw or KW))
    return eng


def _collect(eng, losses):
    from tfkaldi_amd import _lib
    out = {"losses": np.array(losses)}
    for l in range(eng.L + 1):
        out["W%d" % l] = eng.get(_lib.WEIGHTS, l)
    for l in range(eng.L):
        out["beta%d" % l] = eng.get(_lib.BN_BETA, l)
        out["mm%d" % l] = eng.get(_lib.BN_MOVING_MEAN, l)
        out["mv%d" % l] = eng.get(_lib.BN_MOVING_VAR, l)
    return out


def _grads(eng):
    from tfkaldi_amd import _lib
    g = {}
    for l in range(eng.L + 1):
        g["gW%d" % l] = eng.get(_lib.WEIGHTS, l, _lib.SLOT_GRAD)
        g["gb%d" % l] = eng.get(_lib.BIASES, l, _lib.SLOT_GRAD)
    for l in range(eng.L):
        g["gbeta%d" % l] = eng.get(_lib.BN_BETA, l, _lib.SLOT_GRAD)
    return g


def _worker(rank, world, port, num_mb, out_dir, dtype="float32", mode="sharded", kw=None, frames=None, transport="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), TFK_DP_MIN_SHARD="64")
    if transport == "gloo":
        os.environ.update(TFK_SHARE_DEVICE="1", TFK_DIST_BACKEND="gloo", TFK_DP_EMULATE_RS="1")
    else:
        # REAL RCCL: the in-library exchange (csrc/exchange.hip) -- "rccl-torch": the library refuses to load RCCL, every rank
        # agrees on the torch.distributed driver (BucketReducer over the nccl backend); "rccl-mixed": ONE rank cannot load it --
        # the others must hear of that before they enter the collective ncclCommInitRank (tfk_comm_available, MIN-reduced).  On a box with fewer GPUs than ranks the
        # ranks share a device and claim a host each (TFK_FAKE_NODES; dataparallel._share_device)
        os.environ.update(HSA_ENABLE_IPC_MODE_LEGACY="0", TFK_FAKE_NODES=transport.split(":")[1])
        if transport.startswith("rccl-torch") or (transport.startswith("rccl-mixed") and rank == 1):
            os.environ["TFK_RCCL_LIB"] = "/nonexistent/librccl.so"
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from tfkaldi_amd.dataparallel import DataParallel, init_from_env, partition
    init_from_env()
    dp = DataParallel(mode=mode)
    assert dp.enabled
    assert dist.get_backend() == ("gloo" if transport == "gloo" else "nccl")
    eng = _engine(torch_state=True, dtype=dtype, kw=kw)
    data = (lambda n, seed: _data(n, seed)) if frames is None else (lambda n, seed: _data_kw(n, seed, kw, frames))
    extra = {}
    if mode == "allreduce":
        # the reduced gradient SUM itself, before Adam amplifies its rounding noise: one step by hand through the same
        # reducer (bucket callbacks -> collectives -> wait), G read back, then the optimiser
        red = dp.reducer(eng)
        mbs = data(num_mb, 100)
        start, end = partition(len(mbs), world)[rank]
        eng.set_later_microbatches(len(mbs) - end)
        if getattr(red, "native", False):  # the library sits behind the engine's hooks itself
            for i, (X, y) in enumerate(mbs[start:end]):
                eng.accumulate(X, y, last=(i == end - start - 1))
            if start == end:
                red.idle(eng)
            red.finish_reduce()
            extra = _grads(eng)
            extra["loss100"] = np.array(red.finish_and_apply(eng))
        else:
            eng.set_bucket_callback(red.on_bucket)
            for i, (X, y) in enumerate(mbs[start:end]):
                eng.accumulate(X, y, last=(i == end - start - 1))
            if start == end:
                eng.zero_accumulators()
                for b in eng.bucket_order():
                    red.on_bucket(b)
            eng.set_bucket_callback(None)
            red.finish()
            extra = _grads(eng)
            extra["loss100"] = np.array(eng.apply())
    losses = [dp.train_step(eng, data(num_mb, step)) for step in range(3)]
    if mode == "sharded" and num_mb >= 1:
        assert "rs" in dp.last_kinds, dp.last_kinds  # the sharded protocol really ran (reduce-scatter emulated)
        assert getattr(dp.reducer(eng), "verify_left", 0) == 0
    losses.append(dp.eval_step(eng, data(num_mb, 9)))
    losses.append(dp.train_step(eng, data(num_mb, 5)))  # (consumes the gathers that crossed the evaluation)
    red = dp.reducer(eng)
    assert bool(getattr(red, "native", False)) == transport.startswith("rccl:"), (transport, type(red).__name__)
    stale = red.masters_stale
    if stale:  # mixed precision: the fp32 masters are sharded -- reading them must be refused until they are gathered
        try:
            eng.get(0, 0)
            raise AssertionError("stale fp32 masters were handed out")
        except RuntimeError as exc:
            assert "gather_parameters" in str(exc)
    dp.gather_parameters(eng)
    out = _collect(eng, losses)
    out.update(extra)
    out["stale"] = np.array(stale)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
    eng.close()
    dist.destroy_process_group()


def _data_kw(num_mb, seed, kw, frames):
    rng = np.random.default_rng(seed)
    return [((rng.standard_normal((frames + 8 * i, kw["input_dim"]))).astype(np.float32),
             rng.integers(0, kw["output_dim"], size=frames + 8 * i).astype(np.int32)) for i in range(num_mb)]


def _serial(num_mb, dtype, kw, mode, frames=None):
    data = (lambda n, seed: _data(n, seed)) if frames is None else (lambda n, seed: _data_kw(n, seed, kw, frames))
    eng = _engine(torch_state=False, dtype=dtype, kw=kw)
    extra = {}

    def train(seed):
        mbs = data(num_mb, seed)
        for i, (X, y) in enumerate(mbs):
            eng.accumulate(X, y, last=(i == len(mbs) - 1))

    if mode == "allreduce":
        train(100)
        extra = _grads(eng)
        extra["loss100"] = np.array(eng.apply())
    want = []
    for step in range(3):
        train(step)
        want.append(eng.apply())
    for X, y in data(num_mb, 9):
        eng.eval_accumulate(X, y)
    want.append(eng.eval_finish())
    train(5)
    want.append(eng.apply())
    ref = _collect(eng, want)
    ref.update(extra)
    eng.close()
    return ref


def _compare(tmp_path, world, ref, lr, steps, loss_rtol=3e-6, flip_threshold=0.02, stat_tol=(1e-5, 1e-7)):
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert np.allclose(got["losses"], ref["losses"], rtol=loss_rtol, atol=0), (got["losses"], ref["losses"])
        for k in ref:
            if k == "losses":
                continue
            if k.startswith("g") or k == "loss100":
                # the SUM all-reduce of the per-rank gradient sums against the serial accumulation: same addends, another
                # order -- rtol 1e-5 on the scale of the tensor
                scale = np.abs(ref[k]).max() + 1e-30
                assert np.abs(got[k] - ref[k]).max() <= 1e-5 * scale, (k, np.abs(got[k] - ref[k]).max(), scale)
            elif k.startswith("m"):
                assert np.allclose(got[k], ref[k], rtol=stat_tol[0], atol=stat_tol[1]), k
            else:  # Adam amplifies summation-order noise of near-zero gradients: see test_gpu_engine_parity
                err = np.abs(got[k] - ref[k])
                assert np.mean(err > flip_threshold * lr * steps) < 0.01 and err.max() <= 2 * lr * steps, (
                    k, float(np.mean(err > flip_threshold * lr * steps)), float(err.max()))


LAYERWISE = dict(KW, num_layers=3, layerwise_init=True)  # one active hidden layer of three at initialisedlayers = 0


_CASES = [
    (9, "float32", "sharded", None), (3, "float32", "sharded", None), (1, "float32", "sharded", None),
    (3, "bfloat16", "sharded", None), (3, "float32", "allreduce", None), (3, "bfloat16", "allreduce", None),
    # layer-wise growth below full depth with idle ranks: every rank must announce -- and launch -- the same
    # collectives in the same order whatever the active depth is (round-1 advisor finding)
    (1, "float32", "sharded", LAYERWISE), (1, "float32", "allreduce", LAYERWISE), (3, "bfloat16", "sharded", LAYERWISE)]


@pytest.mark.parametrize("world,num_mb,dtype,mode,kw", [(2,) + c for c in _CASES] + [
    (4, 9, "float32", "sharded", None), (4, 3, "bfloat16", "sharded", LAYERWISE), (4, 3, "float32", "allreduce", None),
    (8, 9, "float32", "sharded", None), (8, 3, "bfloat16", "sharded", None), (8, 1, "float32", "allreduce", LAYERWISE)])
def test_ranks_match_serial(gpu, tmp_path, world, num_mb, dtype, mode, kw):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(world, port, num_mb, str(tmp_path), dtype, mode, kw), nprocs=world, join=True)
    ref = _serial(num_mb, dtype, kw, mode)
    _compare(tmp_path, world, ref, KW["init_learning_rate"], 5)
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert bool(got["stale"]) == (dtype == "bfloat16" and mode == "sharded")


def _rccl_transport(world, driver="rccl"):
    """one rank per GPU when the box has them, else every rank on GPU 0 claiming a host of its own"""
    import torch
    return "%s:%d" % (driver, 1 if torch.cuda.device_count() < world else 0)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,num_mb,dtype,mode,kw,driver", [
    (2, 9, "float32", "sharded", None, "rccl"), (2, 3, "bfloat16", "sharded", None, "rccl"), (2, 3, "bfloat16", "allreduce", None, "rccl"),
    (2, 1, "float32", "sharded", LAYERWISE, "rccl"), (2, 3, "float32_mfma", "sharded", None, "rccl"),
    (4, 3, "bfloat16", "sharded", LAYERWISE, "rccl"), (8, 9, "float32", "sharded", None, "rccl"), (8, 3, "bfloat16", "sharded", None, "rccl"),
    (8, 1, "float32", "allreduce", LAYERWISE, "rccl"),
    # the in-library exchange cannot load RCCL: every rank runs the torch.distributed driver over the nccl backend
    (2, 3, "float32", "sharded", None, "rccl-torch"), (2, 3, "bfloat16", "sharded", None, "rccl-torch"),
    (4, 5, "float32", "allreduce", None, "rccl-torch"), (4, 5, "float32", "sharded", None, "rccl-mixed")])
def test_ranks_match_serial_over_real_rccl(gpu, tmp_path, world, num_mb, dtype, mode, kw, driver):
    """test_ranks_match_serial with the ranks talking through REAL RCCL: reduce-scatter / all-gather / all-reduce launched by the
    library at world 2 / 4 / 8 in every arithmetic (bf16: shadow gathers, fp32 masters left with their owners), idle ranks under
    layer-wise growth -- and the torch.distributed driver over the nccl backend when the library cannot bind RCCL"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(world, port, num_mb, str(tmp_path), dtype, mode, kw, None, _rccl_transport(world, driver)), nprocs=world,
             join=True)
    ref = _serial(num_mb, dtype, kw, mode)
    # (moving averages composed from the ranks' increments against one update after the other: a few ulps of the largest term --
    # 2.2e-7 measured at world 4 with 5 micro-batches on every transport, gloo included)
    _compare(tmp_path, world, ref, KW["init_learning_rate"], 5, stat_tol=(1e-5, 5e-7))
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert bool(got["stale"]) == (dtype == "bfloat16" and mode == "sharded")


CFG2 = dict(input_dim=440, num_layers=6, num_units=2048, output_dim=2000, nonlin="relu", batch_norm=True,
            init_learning_rate=1e-3, num_steps=10)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("dtype,transport", [("float32", "rccl"), ("bfloat16", "rccl")])
def test_eight_ranks_at_cfg2_size(gpu, tmp_path, dtype, transport):
    """BASELINE cfg2's network on eight ranks: the real span sizes (3.6 / 16.8 / 16.4 MB), 32 MiB coalescing, shards of
    n / 8, an idle rank (seven micro-batches), real engines, the sharded protocol end to end -- over gloo with the torch driver,
    and over REAL RCCL with the in-library exchange"""
    import torch.multiprocessing as mp
    world, num_mb = 8, 7
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    # (the driver's GPU tier gives the suite 20 minutes: the torch.distributed driver over gloo meets this size in
    #  test_bench_eight_ranks_dry_run -- rounds 3-6 also ran this test over gloo, 40-56 s per arithmetic -- and eight ranks on the
    #  small nets above)
    transport = "gloo" if transport.startswith("gloo") else _rccl_transport(world)
    mp.spawn(_worker, args=(world, port, num_mb, str(tmp_path), dtype, "sharded", CFG2, 96, transport), nprocs=world, join=True)
    ref = _serial(num_mb, dtype, CFG2, "sharded", frames=96)
    # (26 M parameters, generic starting point: after a few Adam steps the summation order of eight partial gradient sums
    # shows in the fifth digit of the loss, as the single-GPU loss traces do against float64 -- profiles/r03_loss_trace_f64.json)
    # the parameters: an element whose gradient sits at round-off level may take its first Adam steps (~lr * sign(g)) the other
    # way; over five steps and 26 M parameters these are more numerous than in the toy nets above -- still under 1 % of the
    # elements beyond a fifth of the distance five full steps cover, none beyond twice that distance
    _compare(tmp_path, world, ref, CFG2["init_learning_rate"], 5, loss_rtol=1e-4 if dtype == "float32" else 2e-3,
             flip_threshold=0.2,  # (BN moving statistics follow the drifting parameters; bf16 operands drift further)
             # (the order in which gloo's ring adds the eight partial sums depends on how a span is chunked: with the 64 MiB
             # spans a moving mean missed round 3's absolute bound of 1e-5, with the 24 MB spans none did)
             stat_tol=(1e-3, 3e-5) if dtype == "float32" else (2e-2, 5e-4))
