"""The exchange step INSIDE the library (csrc/exchange.hip, include/tfkaldi_hip.h: tfk_comm) at world 2 / 4 / 8 on the
one GPU of the test box: the ranks are THREADS of this process, each driving its own engine through the product's
DataParallel + NativeExchange; the group is the library's loopback backend (rendezvous + plain kernels instead of RCCL,
which refuses two ranks per device) -- everything else is the code an 8-GPU job runs: bucket announcements coalesced
into spans, in-place reduce-scatter, Adam on the rank's 1/world of every span, in-place all-gather of the parameters
(of the bf16 shadow in mixed precision) consumed layer by layer by the next forward pass, sharded fp32 masters and
their gather, the replica checksum, idle ranks, evaluation.  Must reproduce the single-engine run that processes all
micro-batches serially (KAT 8c-3; reference neuralNetworks/trainer.py:165-184).  With one RCCL rank the same class
runs over real RCCL: tests/test_gpu_rccl_single_rank.py."""
import ctypes
import os
import sys
import threading

import numpy as np
import pytest

from test_gpu_dp_two_ranks import KW, _collect, _data, _engine

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Group(object):
    """a loopback group + one (engine, DataParallel, NativeExchange) per rank"""

    def __init__(self, world, mode, dtype="float32", kw=None, min_bytes=1 << 12, algo=None, wire=None, planes=False):
        from tfkaldi_amd import _lib
        from tfkaldi_amd.dataparallel import DataParallel, NativeExchange
        self.lib = _lib.load()
        self.world = world
        self.handle = ctypes.c_void_p()
        _lib.check(self.lib.tfk_loopback_create(world, ctypes.byref(self.handle)))
        self.engines, self.dps = [], []
        for rank in range(world):
            eng = _engine(torch_state=False, dtype=dtype, kw=kw)
            dp = DataParallel(mode=mode)
            dp.rank, dp.world, dp._forced = rank, world, True
            dp._reducers[eng] = NativeExchange(eng, mode=mode, min_bytes=min_bytes, loopback=(self.handle, rank))
            if algo is not None or wire is not None:
                dp._reducers[eng].set_exchange(algo, wire)
                assert dp._reducers[eng].exchange_info()["reduce_scatter"] == (algo or "rccl")
            if planes:
                dp._reducers[eng].set_gather(True)
            self.engines.append(eng)
            self.dps.append(dp)

    def run(self, fn):
        """fn(rank, engine, dp) on every rank, each in its own thread (a collective blocks until every rank has posted it)"""
        out, errors = [None] * self.world, []

        def body(rank):
            try:
                out[rank] = fn(rank, self.engines[rank], self.dps[rank])
            except BaseException as exc:  # noqa: BLE001
                errors.append((rank, exc))

        threads = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not any(t.is_alive() for t in threads), "a rank is stuck in a collective"
        if errors:
            raise errors[0][1]
        return out

    def close(self):
        from tfkaldi_amd import _lib
        for eng in self.engines:
            eng.close()  # (closes its comm first)
        _lib.check(self.lib.tfk_loopback_destroy(self.handle))


def _serial(num_mb, dtype="float32", kw=None, data=_data):
    eng = _engine(torch_state=False, dtype=dtype, kw=kw)
    losses = []
    for step in (0, 1, 2):
        mbs = data(num_mb, step)
        for i, (X, y) in enumerate(mbs):
            eng.accumulate(X, y, last=(i == len(mbs) - 1))
        losses.append(eng.apply())
    for X, y in data(num_mb, 9):
        eng.eval_accumulate(X, y)
    losses.append(eng.eval_finish())
    mbs = data(num_mb, 5)
    for i, (X, y) in enumerate(mbs):
        eng.accumulate(X, y, last=(i == len(mbs) - 1))
    losses.append(eng.apply())
    out = _collect(eng, losses)
    eng.close()
    return out


def _rank_program(num_mb, data=_data):
    def program(rank, eng, dp):
        losses = [dp.train_step(eng, data(num_mb, step)) for step in (0, 1, 2)]
        red = dp.reducer(eng)
        assert red.native and red.backend == "loopback"
        info = dict(executed=list(dp.last_executed), spans=list(dp.last_collectives))
        losses.append(dp.eval_step(eng, data(num_mb, 9)))
        losses.append(dp.train_step(eng, data(num_mb, 5)))  # (consumes the gathers that crossed the evaluation)
        info["stale"] = red.masters_stale
        if info["stale"]:  # mixed precision: reading sharded fp32 masters must be refused until they are gathered
            with pytest.raises(RuntimeError, match="gather_parameters"):
                eng.get(0, 0)
        dp.gather_parameters(eng)
        assert not red.masters_stale
        return _collect(eng, losses), info
    return program


def _compare(ref, got, loss_rtol, param_atol, tag):
    assert np.allclose(got["losses"], ref["losses"], rtol=loss_rtol, atol=0), (tag, got["losses"], ref["losses"])
    for k, v in ref.items():
        if k == "losses":
            continue
        err = np.abs(got[k].astype(np.float64) - v.astype(np.float64)).max()
        assert err <= param_atol, (tag, k, err)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode,algo", [("sharded", None), ("sharded", "direct"), ("allreduce", None)])
@pytest.mark.parametrize("world,num_mb", [(2, 2), (2, 5), (4, 4), (4, 3), (8, 8), (8, 3)])
def test_loopback_ranks_equal_the_serial_run_fp32(gpu, world, num_mb, mode, algo):
    """even shards, uneven shards, idle ranks (fewer micro-batches than ranks); every rank ends with the same parameters,
    equal to the serial run's up to the fp32 summation order of G (ranks add their sums, the serial run adds micro-batch
    after micro-batch), which Adam amplifies: bounds as tests/test_gpu_dp_two_ranks.py.  algo "direct": the reduce-scatter as
    sub-span transfers to / from every peer + the owner's rank-ordered sum, the gather as the same movement backwards
    (TFK_DP_ALGO=direct, csrc/exchange.hip) instead of the backend's own collectives"""
    os.environ["TFK_DP_MIN_SHARD"] = "64"
    group = _Group(world, mode, algo=algo)
    try:
        results = group.run(_rank_program(num_mb))
    finally:
        group.close()
    ref = _serial(num_mb)
    for rank, (got, info) in enumerate(results):
        _compare(ref, got, 2e-5, 2e-4, "rank %d" % rank)
        for k, v in results[0][0].items():  # replicas: bit-identical to each other
            np.testing.assert_array_equal(got[k], v, err_msg="rank %d %s" % (rank, k))
        if mode == "sharded":
            assert any("reduce_scatter" in n for n in info["executed"]) and any("all_gather" in n for n in info["executed"])
        else:
            assert set(info["executed"]) == {"loopback:all_reduce"}
        assert not info["stale"]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode,algo,planes", [("sharded", None, False), ("sharded", "direct", False), ("allreduce", None, False),
                                              ("sharded", None, True), ("sharded", "direct", True)])
@pytest.mark.parametrize("world,num_mb", [(2, 2), (4, 3), (8, 8)])
def test_loopback_ranks_equal_the_serial_run_f32x3(gpu, world, num_mb, mode, algo, planes):
    """the fp32-emulating arithmetic under the exchange: a different path from both others -- the sharded mode gathers the fp32
    parameters (the three-plane twins of the weights live outside the arena) and every rank REBUILDS its twins from them before
    the next forward pass (engine.hip: refresh_shadow), the all-reduce mode updates them with the full Adam step.  Same bounds as
    fp32 (it claims to be fp32), replicas bit-identical, and no stale masters: nothing is left sharded in this mode.
    planes (TFK_DP_GATHER=planes / tfk_comm_set_gather): the sharding unit is the weight matrix, the twin ROWS the owner's Adam wrote
    travel (6 B per weight) and nothing is rebuilt -- except for a matrix whose rows do not divide by 2 x world (40 input rows at
    world 8), which keeps the fp32 gather + rebuild; the masters of the others stay with their owners until gather_parameters"""
    os.environ["TFK_DP_MIN_SHARD"] = "64"
    group = _Group(world, mode, dtype="float32x3", algo=algo, planes=planes)
    try:
        results = group.run(_rank_program(num_mb))
    finally:
        group.close()
    ref = _serial(num_mb, dtype="float32x3")
    for rank, (got, info) in enumerate(results):
        _compare(ref, got, 2e-5, 2e-4, "rank %d" % rank)
        for k, v in results[0][0].items():
            np.testing.assert_array_equal(got[k], v, err_msg="rank %d %s" % (rank, k))
        if mode == "sharded":
            assert any("reduce_scatter" in n for n in info["executed"]) and any("all_gather" in n for n in info["executed"])
            assert not any("shadow" in n for n in info["executed"])  # fp32 parameters or twin rows on the wire, not a bf16 shadow
            assert any("three-plane twins" in n for n in info["executed"]) == planes
        assert info["stale"] == planes


@pytest.mark.timeout(600)
@pytest.mark.parametrize("algo", [None, "direct"])
@pytest.mark.parametrize("world,num_mb", [(2, 2), (4, 6), (8, 8), (8, 2)])
def test_loopback_ranks_mixed_precision_sharded_masters(gpu, world, num_mb, algo):
    """bf16 GEMMs: what travels back is the bf16 SHADOW (2 B per parameter); the fp32 masters of a span stay on the rank
    that owns it until gather_parameters -- afterwards every rank holds the same masters, close to the serial run's"""
    os.environ["TFK_DP_MIN_SHARD"] = "64"
    kw = dict(KW, num_units=64, output_dim=24)  # (every leading dimension a multiple of 8: the shadow mirrors the arena)
    group = _Group(world, "sharded", dtype="bfloat16", kw=kw, algo=algo)
    try:
        results = group.run(_rank_program(num_mb))
    finally:
        group.close()
    ref = _serial(num_mb, dtype="bfloat16", kw=kw)
    for rank, (got, info) in enumerate(results):
        assert info["stale"], "the bf16 sharded exchange leaves the masters with their owners"
        assert any("bf16 shadow" in n for n in info["executed"]), info["executed"]
        _compare(ref, got, 2e-3, 2e-3, "rank %d" % rank)
        for k, v in results[0][0].items():
            np.testing.assert_array_equal(got[k], v, err_msg="rank %d %s" % (rank, k))


def _data_cfg2(num_mb, seed):
    rng = np.random.default_rng(1000 + seed)
    return [((rng.standard_normal((256, 440)) * 1.5).astype(np.float32),
             rng.integers(0, 2000, size=256).astype(np.int32)) for _ in range(num_mb)]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("algo,dtype,planes", [(None, "float32", False), ("direct", "float32", False), ("direct", "float32x3", True)])
def test_eight_loopback_ranks_at_cfg2_size(gpu, algo, dtype, planes):
    """BASELINE cfg2's network (26 M parameters, 16 MB hidden-layer spans, the default 32 MiB coalescing): the spans an
    8-GPU job exchanges -- [scalar tail], W6..W2, W1 + W0, [vectors] -- every one dividing by 4 x 8"""
    os.environ.pop("TFK_DP_MIN_SHARD", None)
    kw = dict(input_dim=440, num_layers=6, num_units=2048, output_dim=2000, nonlin="relu", batch_norm=True,
              init_learning_rate=1e-3, num_steps=10, max_frames=256)
    group = _Group(8, "sharded", kw=kw, min_bytes=None, algo=algo, dtype=dtype, planes=planes)
    try:
        results = group.run(_rank_program(8, data=_data_cfg2))
    finally:
        group.close()
    ref = _serial(8, kw=kw, data=_data_cfg2, dtype=dtype)
    # (planes: six of cfg2's seven matrices have 2048 rows = 8 x 256 and travel as twin rows; the 440 input rows do not divide by 16)
    assert results[0][1]["stale"] == planes
    spans = results[0][1]["spans"]
    assert [n for _, n in spans if n > 1 << 20] == [12484608, 8388608, 5095424]  # (as bench.py's line reports them)
    assert sum(n.startswith("loopback:reduce_scatter") for n in results[0][1]["executed"]) == 3
    assert all(("[direct]" in n) == (algo == "direct") for n in results[0][1]["executed"] if "all_reduce" not in n)
    for rank, (got, info) in enumerate(results):
        _compare(ref, got, 2e-5, 5e-4, "rank %d" % rank)
        for k, v in results[0][0].items():
            np.testing.assert_array_equal(got[k], v, err_msg="rank %d %s" % (rank, k))


def test_mismatched_collectives_fail_loudly(gpu):
    """two ranks that launch DIFFERENT collectives (here: different exchange modes) must not hang or mix data"""
    from tfkaldi_amd import _lib
    from tfkaldi_amd.dataparallel import DataParallel, NativeExchange
    os.environ["TFK_DP_MIN_SHARD"] = "64"
    lib = _lib.load()
    handle = ctypes.c_void_p()
    _lib.check(lib.tfk_loopback_create(2, ctypes.byref(handle)))
    engines = [_engine(torch_state=False) for _ in range(2)]
    reds = [NativeExchange(engines[r], mode=("sharded", "allreduce")[r], min_bytes=1 << 12, loopback=(handle, r))
            for r in range(2)]
    errors = []

    def body(rank):
        dp = DataParallel()
        dp.rank, dp.world, dp._forced = rank, 2, True
        dp._reducers[engines[rank]] = reds[rank]
        try:
            dp.train_step(engines[rank], _data(2, 0))
        except Exception as exc:  # noqa: BLE001
            errors.append(str(exc))

    threads = [threading.Thread(target=body, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads)
    assert errors and any("different collectives" in e or "failed earlier" in e for e in errors), errors
    for eng in engines:
        eng.close()
    _lib.check(lib.tfk_loopback_destroy(handle))


def _region(eng):
    """the reduce region [G | scalars | BN increments] on the host"""
    import torch
    ptr, n = eng.reduce_region()
    out = np.empty(n, dtype=np.float32)
    eng.synchronize()
    import ctypes
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    rc = hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), ctypes.c_size_t(4 * n), 2)
    assert rc == 0
    return out


@pytest.mark.parametrize("algo,planes", [(None, False), ("direct", False), (None, True), ("direct", True)])
@pytest.mark.parametrize("dtype", ["float32", "float32x3", "bfloat16"])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_reduce_scattered_shards_hold_the_serial_gradient_sum(gpu, world, dtype, algo, planes):
    """BEFORE Adam (which amplifies round-off): after the collectives of a step, rank r's 1/world of every reduce-scattered
    span and the whole of every all-reduced span hold the gradient sums of the serial run.  One micro-batch per rank: the
    ranks' sums are added in rank order = the order the serial run accumulates micro-batches in, so the sums are
    bit-identical (the scalar tail's BN increments excepted: their closed form is another arithmetic).  With algo "direct" this
    is a property of the PRODUCT's reduce-scatter on real RCCL too (the owner adds in rank order: direct_sum_kernel); RCCL's own
    reduce-scatter promises no order -- the loopback backend's stand-in for it happens to add in rank order as well."""
    os.environ["TFK_DP_MIN_SHARD"] = "64"
    if planes and dtype != "float32x3":
        pytest.skip("owner-written twins exist under the emulated arithmetic only")
    mbs = _data(world, 0)
    serial = _engine(torch_state=False, dtype=dtype)
    for i, (X, y) in enumerate(mbs):
        serial.accumulate(X, y, last=(i == world - 1))
    want = _region(serial)
    num_params = serial.buckets()[-1][0]
    serial.close()
    group = _Group(world, "sharded", dtype=dtype, algo=algo, planes=planes)

    def program(rank, eng, dp):
        red = dp.reducer(eng)
        eng.set_later_microbatches(world - 1 - rank)
        eng.accumulate(*mbs[rank], last=True)
        red.finish_reduce()
        got = _region(eng)
        red.finish_and_apply(eng)
        mine = {span: red.my_shards(*span) for span in red.last_launched}  # (plane gathers: a row block of every matrix of the span)
        return got, list(red.last_launched), list(red.last_span_kinds), mine

    try:
        results = group.run(program)
    finally:
        group.close()
    for rank, (got, spans, kinds, mine) in enumerate(results):
        assert kinds.count("rs") >= 2 and kinds.count("ar") == 2, kinds
        for (off, n), kind in zip(spans, kinds):
            if off >= num_params:  # loss, frames, #micro-batches (exact) + BN increments (closed form: close)
                np.testing.assert_array_equal(got[off + 1:off + 3], want[off + 1:off + 3])
                assert np.allclose(got[off:off + n], want[off:off + n], rtol=1e-5, atol=1e-7)
                continue
            parts = [slice(o, o + m) for o, m in mine[(off, n)]] if kind == "rs" else [slice(off, off + n)]
            assert sum(p.stop - p.start for p in parts) == (n // world if kind == "rs" else n)
            for part in parts:
                np.testing.assert_array_equal(got[part], want[part], err_msg="rank %d %s span %s" % (rank, kind, (off, n)))


@pytest.mark.parametrize("dtype,planes", [("float32", False), ("float32x3", False), ("float32x3", True), ("bfloat16", False)])
@pytest.mark.parametrize("world", [2, 8])
def test_bf16_wire_reduce_scatter_against_the_fp32_wire(gpu, world, monkeypatch, dtype, planes):
    """TFK_DP_WIRE=bf16 (off by default): a reduce-scattered span travels as bf16 -- 2 B per parameter in instead of 4 -- every
    rank sends sub-span q to rank q (all-to-all over the point-to-point links) and the OWNER adds the world contributions in fp32,
    in rank order, its own one exact.  Stated tolerance, checked BEFORE Adam on every rank's shard: |G_bf16 - G_serial| <=
    2^-8 * sum over the other ranks of |g_q| (each travelled value is off by at most half a bf16 ulp = 2^-9 relative; factor 2 of
    slack) + the fp32 order noise of the serial sum.  All-reduced spans (vectors, scalar tail) keep fp32.  Then three steps
    end to end: losses within 2e-3 of the fp32 wire's, replicas bit-identical to each other.  In every arithmetic (the wire format
    is independent of what the contractions compute in), and with plane gathers (per-matrix shards, each with its own staging)."""
    os.environ["TFK_DP_MIN_SHARD"] = "64"
    mbs = _data(world, 0)
    serial = _engine(torch_state=False, dtype=dtype)
    contrib, prev = [], None
    for i, (X, y) in enumerate(mbs):  # G after every micro-batch: the differences are the ranks' own contributions
        serial.accumulate(X, y, last=(i == world - 1))
        cur = _region(serial).astype(np.float64)
        contrib.append(cur if prev is None else cur - prev)
        prev = cur
    want = prev
    num_params = serial.buckets()[-1][0]
    serial.close()
    monkeypatch.setenv("TFK_DP_WIRE", "bf16")
    group = _Group(world, "sharded", dtype=dtype, planes=planes)

    def program(rank, eng, dp):
        red = dp.reducer(eng)
        eng.set_later_microbatches(world - 1 - rank)
        eng.accumulate(*mbs[rank], last=True)
        red.finish_reduce()
        got = _region(eng)
        red.finish_and_apply(eng)
        return got, list(red.last_launched), list(red.last_span_kinds), {span: red.my_shards(*span) for span in red.last_launched}

    try:
        results = group.run(program)
    finally:
        group.close()
    seen_rs = 0
    for rank, (got, spans, kinds, mine) in enumerate(results):
        others = sum(np.abs(c) for q, c in enumerate(contrib) if q != rank)
        for (off, n), kind in zip(spans, kinds):
            if off >= num_params:
                continue
            if kind == "rs":
                seen_rs += 1
                worst = 0.0
                for o, m in mine[(off, n)]:
                    part = slice(o, o + m)
                    err = np.abs(got[part].astype(np.float64) - want[part])
                    bound = 2.0 ** -8 * others[part] + 1e-6 * np.abs(want[part]) + 1e-12
                    assert (err <= bound).all(), "rank %d span %s: %.3g x the bound" % (rank, (off, n), float((err / bound).max()))
                    worst = max(worst, float(err.max()))
                assert worst > 0.0  # (it really travelled in bf16)
            else:  # all-reduced spans keep the fp32 wire: bit-identical to the serial sums
                np.testing.assert_array_equal(got[off:off + n], want[off:off + n].astype(np.float32))
    assert seen_rs >= 2 * world
    # end to end against the fp32 wire
    group = _Group(world, "sharded", dtype=dtype, planes=planes)
    try:
        bf = group.run(_rank_program(world))
    finally:
        group.close()
    monkeypatch.delenv("TFK_DP_WIRE")
    group = _Group(world, "sharded", dtype=dtype, planes=planes)
    try:
        fp = group.run(_rank_program(world))
    finally:
        group.close()
    for rank in range(world):
        assert np.allclose(bf[rank][0]["losses"], fp[rank][0]["losses"], rtol=2e-3, atol=0), (bf[rank][0]["losses"], fp[rank][0]["losses"])
        for k, v in bf[0][0].items():
            np.testing.assert_array_equal(bf[rank][0][k], v, err_msg="rank %d %s" % (rank, k))


def test_exchange_options_tuning_and_phase_times(gpu):
    """ABI 8: tfk_comm_set_exchange is refused in the middle of a step and takes effect between steps; tfk_comm_tune (what a
    real RCCL group of more than one rank runs at attach under TFK_DP_ALGO=auto) times both algorithms on scratch memory, agrees
    on the slowest rank's figures and leaves the same choice on every rank; tfk_comm_timing brackets the phases of the step with
    timing events: three steps later every phase that must have run has a positive device time and the step count is three"""
    os.environ["TFK_DP_MIN_SHARD"] = "64"
    world = 4
    group = _Group(world, "sharded", dtype="float32x3")

    def program(rank, eng, dp):
        red = dp.reducer(eng)
        info0 = red.exchange_info()
        assert info0 == {"reduce_scatter": "rccl", "all_gather": "rccl", "wire": "fp32", "chosen_by": "default"}, info0
        tuned = red.tune(1 << 20, iters=3)
        assert tuned["chosen_by"] == "tuned at attach" and all(v > 0 for v in tuned["tuned_us_slowest_rank"].values()), tuned
        losses = [dp.train_step(eng, _data(world, 0))]
        red.set_exchange("direct", "bf16")
        losses.append(dp.train_step(eng, _data(world, 1)))
        assert red.exchange_info()["wire"] == "bf16" and red.exchange_info()["chosen_by"] == "set"
        red.set_exchange("direct", "fp32")
        red.set_gather(True)   # twin rows instead of parameters + rebuild ...
        losses.append(dp.train_step(eng, _data(world, 6)))
        assert red.masters_stale and "three-plane twins" in " ".join(dp.last_executed)
        red.set_gather(False)  # ... and back: the masters come home inside the call (collective)
        assert not red.masters_stale
        losses.append(dp.train_step(eng, _data(world, 7)))
        # refused while a step is in flight: announce a micro-batch, then try
        eng.set_later_microbatches(world - 1 - rank)
        eng.accumulate(*_data(world, 2)[rank], last=True)
        with pytest.raises(RuntimeError, match="middle of a step"):
            red.set_exchange("rccl", None)
        losses.append(red.finish_and_apply(eng))
        red.timing_begin()
        for step in (3, 4, 5):
            losses.append(dp.train_step(eng, _data(world, step)))
        phases, steps = red.timing_read()
        dp.gather_parameters(eng)
        return tuned, phases, steps, losses

    try:
        results = group.run(program)
    finally:
        group.close()
    assert all(r[0] == results[0][0] for r in results), [r[0] for r in results]  # the same figures, the same choice everywhere
    for tuned, phases, steps, losses in results:
        assert steps == 3 and np.isfinite(losses).all()
        assert set(phases) == {"reduce_scatter", "all_reduce", "tail_exposed", "adam", "all_gather", "twin_rebuild", "gather_exposed"}
        for k in ("reduce_scatter", "all_reduce", "tail_exposed", "adam", "all_gather", "twin_rebuild", "gather_exposed"):
            assert phases[k] > 0.0, (k, phases)
        assert all(v < 1e3 for v in phases.values()), phases
    print("tuned (loopback, 4 ranks, 1 Mi floats):", results[0][0])
    print("phases, ms per step (rank 0):", results[0][1])


@pytest.mark.parametrize("dtype,planes", [("bfloat16", False), ("float32x3", True), ("float32x3", "toggle"), ("float32", False)])
def test_span_size_changes_between_steps_with_masters_left_at_their_owners(gpu, dtype, planes):
    """tfk_comm_set_bucket_bytes (bench.py's span-size sweep) cuts the spans differently, i.e. assigns shards to other ranks.  In the
    configurations that leave fp32 masters with their owners between steps (mixed precision: the bf16 shadow travels; emulated
    fp32 with plane gathers) the new owner would update STALE masters unless they are gathered first -- and in EVERY configuration
    Adam's moments are current on a shard's owner only (the optimiser state is sharded with the optimiser: this test caught the
    first version of the call gathering the masters alone).  tfk_comm_set_bucket_bytes / tfk_comm_set_gather bring both home
    (collective).  Who owns an element changes nothing in what is computed for it (Adam is element-wise, the loopback group adds
    in rank order whatever the cut), so six steps over three span sizes -- "toggle": and over parameter gathers / plane gathers /
    parameter gathers -- must equal six steps at ONE span size BIT FOR BIT, and the serial run at the arithmetic's usual tolerance."""
    os.environ["TFK_DP_MIN_SHARD"] = "64"
    world = 4
    kw = dict(KW, num_units=64, output_dim=24)

    stale = dtype == "bfloat16" or planes is True

    def run(sizes, toggle=False):
        group = _Group(world, "sharded", dtype=dtype, kw=kw, planes=planes is True)

        def program(rank, eng, dp):
            red = dp.reducer(eng)
            losses, cuts = [], []
            for k, size in enumerate(sizes):
                red.set_bucket_bytes(size)
                if toggle:
                    red.set_gather(k == 1)
                for step in (2 * k, 2 * k + 1):
                    losses.append(dp.train_step(eng, _data(world, step)))
                assert red.masters_stale == (stale or (toggle and k == 1))
                cuts.append(tuple(n for _, n in red.last_launched))
            dp.gather_parameters(eng)
            return _collect(eng, losses), cuts

        try:
            return group.run(program)
        finally:
            group.close()

    changing = run((1 << 12, 1 << 22, 1 << 14), toggle=planes == "toggle")
    fixed = run((1 << 12, 1 << 12, 1 << 12))
    assert len(set(changing[0][1])) >= 2, changing[0][1]  # the spans really were cut differently
    assert len(set(fixed[0][1])) == 1
    for rank in range(world):
        for k, v in fixed[0][0].items():
            np.testing.assert_array_equal(changing[rank][0][k], v, err_msg="rank %d %s" % (rank, k))
    serial = _engine(torch_state=False, dtype=dtype, kw=kw)
    want = []
    for step in range(6):
        mbs = _data(world, step)
        for i, (X, y) in enumerate(mbs):
            serial.accumulate(X, y, last=(i == len(mbs) - 1))
        want.append(serial.apply())
    ref = _collect(serial, want)
    serial.close()
    tol = (2e-3, 5e-3) if dtype == "bfloat16" else (2e-5, 4e-4)  # (six Adam steps: a little more room than the four-step tests)
    _compare(ref, changing[0][0], tol[0], tol[1], "rank 0")


@pytest.mark.parametrize("dtype", ["float32x3", "bfloat16"])
def test_no_exchange_option_changes_what_is_computed(gpu, dtype):
    """Algorithm, gather, span size, the tuning pass and the phase timing are about HOW the sums travel: on the fp32 wire none of
    them may change a bit of the result (the loopback group adds in rank order under either algorithm).  Eight steps, an option
    flipped before every one, against eight steps with everything at its default."""
    os.environ["TFK_DP_MIN_SHARD"] = "64"
    world = 4
    kw = dict(KW, num_units=64, output_dim=24)
    x3 = dtype == "float32x3"

    def run(flip):
        group = _Group(world, "sharded", dtype=dtype, kw=kw)

        def program(rank, eng, dp):
            red = dp.reducer(eng)
            flips = [lambda: red.set_exchange("direct", None), lambda: red.tune(1 << 16, iters=2), lambda: red.timing_begin(),
                     lambda: red.set_gather(True) if x3 else red.set_bucket_bytes(1 << 20), lambda: red.set_exchange("rccl", None),
                     lambda: red.timing_read(), lambda: red.set_gather(False) if x3 else red.set_bucket_bytes(1 << 12),
                     lambda: red.set_exchange("direct", "fp32")]
            losses = []
            for step in range(8):
                if flip:
                    flips[step]()
                losses.append(dp.train_step(eng, _data(world, step)))
            dp.gather_parameters(eng)
            return _collect(eng, losses)

        try:
            return group.run(program)
        finally:
            group.close()

    flipped, plain = run(True), run(False)
    for rank in range(world):
        for k, v in plain[0].items():
            np.testing.assert_array_equal(flipped[rank][k], v, err_msg="rank %d %s" % (rank, k))
