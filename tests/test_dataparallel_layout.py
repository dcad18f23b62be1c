"""The exchange step at 2, 4 and 8 ranks on CPU (gloo) with the bucket layout of BASELINE cfg2 -- 26 M parameters, weight spans
of 3.6 / 16.8 / 16.4 MB, 32 MiB coalescing, the n % (4 * world) shard rule -- driven by the PRODUCT's DataParallel /
BucketReducer over tests/layout_engine.LayoutEngine (exact float32 arithmetic: a sharded run equals the serial run BIT FOR
BIT).  Covers what the driver's `bench.py --gpus 8` meets: both exchange modes, idle ranks (fewer micro-batches than ranks),
layer-wise growth below full depth, asynchronous parameter gathers consumed layer by layer by the next step, the
mixed-precision variant that gathers the bf16 shadow and keeps the fp32 masters sharded, and the replica checksum.
Reference seam: neuralNetworks/trainer.py:165-169 (G += g, loss, frames), :174-184 (mean -> clip -> Adam)."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG2 = dict(F=440, L=6, H=2048, O=2000)
SMALL = dict(F=40, L=3, H=64, O=24)   # (spans of 2560 / 4096 / 1536 floats)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _microbatches(step, num_mb):
    return [(100 * step + i, 8 + (i % 5)) for i in range(num_mb)]


def _run(eng, dp, steps, num_mb):
    losses = []
    for step in range(steps):
        losses.append(dp.train_step(eng, _microbatches(step, num_mb)))
    losses.append(dp.eval_step(eng, _microbatches(9, num_mb)))
    # one more training step after the evaluation: its forward pass is the consumer of the last gathers
    losses.append(dp.train_step(eng, _microbatches(steps, num_mb)))
    return losses


def _worker(rank, world, port, out_dir, shape, mode, bf16, nact, num_mb, min_shard):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS="1")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    from layout_engine import Layout, LayoutEngine
    import tfkaldi_amd.dataparallel as dpmod
    from tfkaldi_amd.dataparallel import DataParallel, init_from_env
    assert init_from_env() == (rank, world, 0)
    if min_shard:
        dpmod.BucketReducer.MIN_SHARD_FLOATS = min_shard
    dp = DataParallel(mode=mode)
    eng = LayoutEngine(Layout(**shape), bf16=bf16, nact=nact)
    losses = _run(eng, dp, 2, num_mb)
    red = dp.reducer(eng)
    info = dict(mode=red.mode, kinds=list(dp.last_kinds), executed=list(dp.last_executed),
                launched=[list(x) for x in dp.last_collectives], stale=bool(red.masters_stale),
                verify_left=red.verify_left, pending=len(red.pending))
    refused = False
    if red.masters_stale:
        try:
            eng.get_params()
        except RuntimeError as exc:
            refused = "gather_parameters" in str(exc)
    dp.gather_parameters(eng)  # collective
    params = eng.get_params()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), losses=np.array(losses), params=params, mov=eng.mov,
             digests=np.array([(k[0], k[1], v) for k, v in sorted(eng.digests.items())], dtype=np.int64).reshape(-1, 3),
             refused=np.array(refused),
             shadow=(eng.shadow.view(torch.int16).numpy() if eng.shadow is not None else np.zeros(0, dtype=np.int16)))
    import json
    json.dump(info, open(os.path.join(out_dir, "rank%d.json" % rank), "w"))
    eng.close()
    assert eng not in dp._reducers
    dist.destroy_process_group()


def _serial(shape, bf16, nact, num_mb):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from layout_engine import Layout, LayoutEngine
    from tfkaldi_amd.dataparallel import DataParallel
    eng = LayoutEngine(Layout(**shape), bf16=bf16, nact=nact)
    dp = DataParallel()
    assert not dp.enabled
    losses = _run(eng, dp, 2, num_mb)
    return eng, losses


def _check(tmp_path, world, shape, mode, bf16, nact, num_mb, min_shard=0):
    import json
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), shape, mode, bf16, nact, num_mb, min_shard),
             nprocs=world, join=True)
    ref, want = _serial(shape, bf16, nact, num_mb)
    infos = []
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        info = json.load(open(os.path.join(str(tmp_path), "rank%d.json" % rank)))
        infos.append(info)
        assert np.array_equal(got["losses"], np.array(want)), (rank, got["losses"], want)
        assert np.array_equal(got["params"], ref.params.numpy()), "rank %d: parameters differ from the serial run" % rank
        assert np.array_equal(got["mov"], ref.mov), "rank %d: BN moving averages" % rank
        # every forward pass read exactly the parameters the serial run read at that point: no gather was still in flight
        seen = {(int(a), int(b)): int(c) for a, b, c in got["digests"]}
        assert seen or num_mb < world  # (a rank that never had a micro-batch never ran a forward pass)
        for key, value in seen.items():
            assert ref.digests[key] == value, "rank %d read stale parameters of layer %d after %d steps" % (rank, key[1], key[0])
        if bf16:
            assert np.array_equal(got["shadow"], ref.shadow.view(__import__("torch").int16).numpy())
    return infos


def test_layout_matches_the_library():
    """the layout restatement of the test double against the library's own arithmetic (tfk_state_bytes; no GPU needed)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from layout_engine import Layout
    from tfkaldi_amd import _lib
    lib = _lib.load()
    for shape, bn in ((CFG2, True), (SMALL, True), (dict(F=440, L=8, H=4096, O=8000), True), (dict(F=30, L=2, H=50, O=7), False),
                      (dict(F=440, L=6, H=2048, O=4000), True)):
        lay = Layout(batch_norm=bn, **shape)
        for dtype in ("float32", "bfloat16"):
            n = ctypes.c_size_t()
            cfg = _lib.make_config(shape["F"], shape["L"], shape["H"], shape["O"], batch_norm=bn, compute_dtype=dtype)
            assert lib.tfk_state_bytes(ctypes.byref(cfg), ctypes.byref(n)) == 0
            assert n.value == lay.state_bytes(dtype == "bfloat16"), (shape, dtype)
    lay = Layout(**CFG2)
    assert lay.P >= 25995216 and len(lay.buckets()) == 9
    assert [n * 4 >> 20 for _, n in lay.buckets()[:7]] == [15, 16, 16, 16, 16, 16, 3]  # MiB: W_6 .. W_0


@pytest.mark.parametrize("world,mode,bf16,nact,num_mb", [
    (8, "sharded", False, None, 8),     # one micro-batch per rank: the driver's `bench.py --gpus 8`
    (8, "sharded", True, None, 5),      # bf16 shadow gathers, three idle ranks
    (8, "allreduce", False, 2, 11),     # uneven blocks, layer-wise growth below full depth
    (4, "sharded", True, 3, 3),         # idle rank + growth + shadow
    (4, "sharded", False, None, 6),
    (2, "sharded", False, None, 4),
])
def test_cfg2_layout_ranks_equal_serial(tmp_path, world, mode, bf16, nact, num_mb):
    infos = _check(tmp_path, world, CFG2, mode, bf16, nact, num_mb)
    lay_w = [(0, 901120), (901120, 4194304), (5095424, 4194304), (9289728, 4194304), (13484032, 4194304),
             (17678336, 4194304), (21872640, 4096000)]
    for info in infos:
        assert info["mode"] == mode and info["verify_left"] == 0 or mode == "allreduce"
        launched = [tuple(x) for x in info["launched"]]
        covered = np.zeros(lay_w[-1][0] + lay_w[-1][1] + 64 * 1024, dtype=np.int8)
        for off, n in launched:
            covered[off:off + n] += 1
        assert (covered[:lay_w[-1][0] + lay_w[-1][1]] == 1).all()  # every gradient reduced exactly once
        if mode == "sharded":
            # the 32 MiB rule: [W_6 .. W_4] (49.9 MB: W_6 + W_5 are 33.2 MB, just short of 32 MiB), [W_3 + W_2] (32 MiB exactly), then
            # [W_1 + W_0] (20.4 MB, flushed at the end)
            rs = [x for x, k in zip(launched, info["kinds"]) if k == "rs"]
            assert rs == [(13484032, 12484608), (5095424, 8388608), (0, 5095424)], rs
            assert all(n % (4 * world) == 0 for _, n in rs)
            assert info["executed"].count("reduce_scatter_tensor") == 3
            assert info["stale"] == bool(bf16)
        else:
            assert set(info["kinds"]) == {"ar"} and set(info["executed"]) == {"all_reduce"}


@pytest.mark.parametrize("world", [2, 4, 8])
def test_small_layout_all_worlds(tmp_path, world):
    """a layout whose spans do NOT all divide by 4 * world: those are all-reduced, the rest sharded; more ranks than
    micro-batches"""
    infos = _check(tmp_path, world, SMALL, "sharded", True, 2, 3, min_shard=64)
    for info in infos:
        assert "rs" in info["kinds"] and "ar" in info["kinds"], info
        assert info["stale"]


def test_sharded_masters_are_refused_until_gathered(tmp_path):
    world = 2
    _check(tmp_path, world, SMALL, "sharded", True, None, 2, min_shard=64)
    for rank in range(world):
        assert bool(np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))["refused"])
