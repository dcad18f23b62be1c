"""The RCCL call path on the one GPU of the test box: a single-rank "nccl" process group (TFK_FORCE_DP=1) drives
DataParallel.train_step / eval_step exactly as a multi-GPU job does -- torch-owned engine state, bucket callback
-> async all-reduce on RCCL's stream -> wait on the engine stream -> apply -- and must reproduce the plain
single-process run bit for bit (a 1-rank SUM all-reduce is the identity).  bench.py's N>1 branch is run the same
way."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_dp_two_ranks import _collect, _data, _engine

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def _worker(rank, port, num_mb, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      TFK_FORCE_DP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from tfkaldi_amd.dataparallel import DataParallel, init_from_env
    init_from_env()
    assert dist.get_backend() == "nccl"
    dp = DataParallel()
    assert dp.enabled
    eng = _engine(torch_state=True)
    losses = [dp.train_step(eng, _data(num_mb, step)) for step in range(3)]
    losses.append(dp.eval_step(eng, _data(num_mb, 9)))
    np.savez(os.path.join(out_dir, "rccl.npz"), **_collect(eng, losses))
    eng.close()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_single_rank_rccl_is_identity(gpu, tmp_path):
    import torch.multiprocessing as mp
    num_mb = 3
    mp.spawn(_worker, args=(_free_port(), num_mb, str(tmp_path)), nprocs=1, join=True)
    eng = _engine(torch_state=False)
    want = []
    for step in range(3):
        mbs = _data(num_mb, step)
        for i, (X, y) in enumerate(mbs):
            eng.accumulate(X, y, last=(i == len(mbs) - 1))
        want.append(eng.apply())
    for X, y in _data(num_mb, 9):
        eng.eval_accumulate(X, y)
    want.append(eng.eval_finish())
    ref = _collect(eng, want)
    eng.close()
    got = np.load(os.path.join(str(tmp_path), "rccl.npz"))
    for k, v in ref.items():
        np.testing.assert_array_equal(got[k], v, err_msg=k)


@pytest.mark.timeout(600)
def test_bench_dp_branch_over_rccl(gpu):
    """bench.py's N>1 code path (callback all-reduce + barrier + MAX over ranks) with one RCCL rank"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", TFK_FORCE_DP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup",
                          "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert abs(line["loss_first_last"][0] - np.log(2000)) < 1e-3
