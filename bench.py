"""Benchmark of the DNN training hot path on MI355X (contract: python bench.py --gpus N --steps K --warmup W).

Workload = BASELINE.json configs[1] ("cfg2"): 6x2048 ReLU + batch-norm DNN, 40-dim fbank +-5 splice = 440
inputs, 2000 pdf-ids, 1024 frames (16 utterances x 64 frames) per GPU per optimiser step, fp32 (exact-fp32 MFMA).
One "step" = one full optimiser step of the reference's Trainer.update: forward + softmax-CE + backward on the
micro-batch, gradient exchange when N > 1, mean -> clip -> Adam, BN moving averages, loss returned to the host.
Weak scaling: every rank owns its own 1024-frame micro-batch (one micro-batch per GPU, as the reference's
128-utterance batch in 16-utterance micro-batches maps onto 8 GPUs); `value` = all ranks' frames / max-rank time.
Inputs come from synthetic ark/scp/alignment files through the product's feature reader + batch dispenser; a ring
of DISTINCT micro-batches (a different one every step) is resident in HBM before the timed region.

`--gpus N` with N > 1 from a bare shell launches N ranks itself (torch.distributed.run, one rank per GPU over RCCL);
under torchrun / the driver's own launcher (WORLD_SIZE set) it joins that group.  Prints ONE JSON line on rank 0.

Besides the contract's fields the line carries: `roofline` (dominant kernel, live HIP-event timing),
`host_fed_value` (the same step fed from HOST numpy through tfk_accumulate: PCIe inclusive, never `value`),
`loss_trace_gpu` / `loss_trace_cpu` (per-step average_loss of the first 20 steps from the engine and from the CPU
stand-in on the same micro-batch sequence) with their largest relative difference, `posterior_max_err` (decoder
posteriors of one utterance vs the float64 oracle holding the engine's parameters), and `cpu_baseline`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_RAW, CONTEXT, L, H, O = 40, 5, 6, 2048, 2000
UTT_PER_GPU, UTT_LEN = 16, 64
F = F_RAW * (2 * CONTEXT + 1)
T = UTT_PER_GPU * UTT_LEN
M_MACS = F * H + (L - 1) * H * H + H * O
FLOP_PER_FRAME = 6 * M_MACS - 2 * F * H  # fwd 2M, dW 2M, dA 2M minus the unneeded layer-0 dA
PEAK_FP32_MFMA_TFLOPS = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0           # v_mfma_f32_32x32x16_bf16, dense (--dtype bfloat16 only)
TRACE_STEPS = 20                         # per-step average_loss of the first 20 steps
MAX_RING = 16                            # distinct micro-batches resident per rank


def self_launch(args):
    """`python bench.py --gpus N` from a bare shell: re-execute under torch.distributed.run, one rank per GPU."""
    import torch
    share = os.environ.get("TFK_SHARE_DEVICE") == "1"  # tests only: several ranks on one GPU (gloo)
    have = torch.cuda.device_count()
    if not share and have < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node" % (args.gpus, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def make_batches(rank, world, count, workdir):
    """`count` micro-batches of this rank through the product I/O path (ark -> CMVN -> splice -> dispenser): step i
    of the job is the reference's batch i (world * 16 utterances in dispenser order), of which rank r takes the
    r-th 16-utterance micro-batch."""
    from tfkaldi_amd import synthetic
    from tfkaldi_amd.processing import batchdispenser, feature_reader, target_coder
    paths = synthetic.write_corpus(workdir, UTT_PER_GPU * world * count, O, feat_dim=F_RAW, utt_len=UTT_LEN)
    reader = feature_reader.FeatureReader(paths["feats_scp"], paths["cmvn_scp"], paths["utt2spk"], CONTEXT, UTT_LEN)
    coder = target_coder.AlignmentCoder(lambda x, y: x, O)
    disp = batchdispenser.AlignmentBatchDispenser(reader, coder, UTT_PER_GPU, paths["alignments"])
    out = []
    for g in range(world * count):
        xs, ys = disp.get_batch()
        if g % world == rank:
            out.append((np.ascontiguousarray(np.concatenate(xs, 0)), np.concatenate(ys, 0).astype(np.int32)))
    return out


def cpu_baseline(batches, hidden_weights, budget_s=20.0):
    """the same optimiser steps on the host cores (PyTorch-CPU restatement; TensorFlow is not installable here)"""
    import torch
    from oracle.torch_cpu_step import TorchCpuTrainer
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 64)  # beyond ~64 threads the 1024-row GEMMs of this step stop scaling on the host
    t = TorchCpuTrainer(F, L, H, O, nonlin="relu", batch_norm=True, threads=cores)
    t.set_hidden_weights(hidden_weights)
    trace, steps, t0 = [], 0, time.perf_counter()
    while True:
        X, y = batches[steps % len(batches)]
        t.accumulate(X, y)
        trace.append(t.apply())
        steps += 1
        dt = time.perf_counter() - t0
        if (dt > budget_s and steps >= TRACE_STEPS) or steps >= 60 or dt > 3 * budget_s:
            break
    return {"value": steps * T / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d optimiser steps from the same initial weights over the same micro-batch sequence (%d frames "
                      "each), no warm-up, PyTorch-CPU fp32 restatement of CrossEnthropyTrainer.update (a stand-in: "
                      "TensorFlow, the reference's CPU path, is absent)" % (steps, T)}, trace[:TRACE_STEPS]


def posterior_error(eng, X):
    """max |engine posterior - float64 oracle posterior| on one utterance, the oracle holding the engine's parameters"""
    from oracle.dnn_oracle import OracleDNN
    from tfkaldi_amd import _lib
    orc = OracleDNN(F, L, H, O, nonlin="relu", batch_norm=True)
    for l in range(L + 1):
        orc.W[l] = eng.get(_lib.WEIGHTS, l).astype(np.float64)
        orc.b[l] = eng.get(_lib.BIASES, l).astype(np.float64)
    for l in range(L):
        orc.beta[l] = eng.get(_lib.BN_BETA, l).astype(np.float64)
        orc.mov_mean[l] = eng.get(_lib.BN_MOVING_MEAN, l).astype(np.float64)
        orc.mov_var[l] = eng.get(_lib.BN_MOVING_VAR, l).astype(np.float64)
    return float(np.abs(eng.posteriors(X).astype(np.float64) - orc.posteriors(X)).max())


def measured_traffic(kernel_name):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC summary (tools/profile_round.sh +
    tools/hbm_traffic.py), valid only for the kernel sources it was measured on."""
    from tfkaldi_amd.build import csrc_hash
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(path):
        return None, "no profiles/hbm_traffic.json"
    rec = json.load(open(path))
    meta = rec.get("_meta", {})
    if meta.get("csrc_sha16") != csrc_hash():
        return None, "profiles/hbm_traffic.json is stale (measured on csrc %s, this build is %s): dropped" % (
            meta.get("csrc_sha16"), csrc_hash())
    if kernel_name not in rec:
        return None, "kernel not in profiles/hbm_traffic.json"
    return rec[kernel_name]["bytes_per_launch"], (
        "profiles/hbm_traffic.json (csrc %s, %s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this "
        "command, (2*FETCH+WRITE)*1024 per MI355X_MICROARCH.md; L2<->fabric side, Infinity-Cache hits included"
        % (meta.get("csrc_sha16"), meta.get("measured", "?")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f64-trace", action="store_true",
                    help="skip the float64-oracle loss trace (about a minute of host numpy behind the CPU baseline)")
    ap.add_argument("--dtype", choices=["float32", "bfloat16"], default="float32",
                    help="float32 (default) is BASELINE cfg2's arithmetic and the only valid headline; bfloat16 runs "
                         "the same workload in the engine's mixed-precision mode (cfg3/cfg4 arithmetic) for reference")
    ap.add_argument("--exchange", choices=["sharded", "allreduce"], default=None,
                    help="N > 1: reduce-scatter + sharded Adam + all-gather (default) or all-reduce + full Adam")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    # stdout carries exactly ONE line, the JSON record: libraries that chat on stdout (RCCL prints its library
    # path from C stdio at teardown) are diverted to stderr until the record is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from tfkaldi_amd import _lib
    from tfkaldi_amd.dataparallel import DataParallel, init_from_env
    from tfkaldi_amd.engine import Engine

    rank, world, local_rank = init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)  # (init_from_env folds local_rank onto one device under TFK_SHARE_DEVICE)
    dp = DataParallel()

    total_steps = args.steps + args.warmup
    ring = max(1, min(total_steps, MAX_RING))
    with tempfile.TemporaryDirectory(prefix="tfkaldi_bench_") as workdir:
        batches = make_batches(rank, world, ring, os.path.join(workdir, "rank%d" % rank))
    assert all(X.shape == (T, F) and X.dtype == np.float32 and y.shape == (T,) for X, y in batches)

    cfg = _lib.make_config(F, L, H, O, nonlin="relu", batch_norm=True, init_learning_rate=1e-3,
                           num_steps=3 * total_steps, max_frames=T, device=local_rank, compute_dtype=args.dtype)
    eng = Engine(cfg, torch_state=dp.enabled)
    rng = np.random.default_rng(7)
    hidden = [(rng.standard_normal((F if l == 0 else H, H)) / np.sqrt(F if l == 0 else H)).astype(np.float32)
              for l in range(L)]
    for l, w in enumerate(hidden):
        eng.set(_lib.WEIGHTS, l, w)

    dev = [(torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()) for X, y in batches]
    torch.cuda.synchronize()

    reducer = None
    if dp.enabled:
        import torch.distributed as dist
        # the product's exchange step (tfkaldi_amd/dataparallel.py): per-layer bucket announcements from backward,
        # coalesced into a few large asynchronous collectives launched while backward is still being enqueued
        dp.mode = args.exchange
        reducer = dp.reducer(eng)  # (collective: probes what the backend can do)
        eng.set_bucket_callback(reducer.on_bucket)
        eng.set_later_microbatches(world - 1 - rank)

    counter = [0]

    def finish():
        if dp.enabled:
            return reducer.finish_and_apply(eng)  # optimiser per reduced span, behind the collectives still in flight
        return eng.apply()

    def step():
        dX, dy = dev[counter[0] % ring]
        counter[0] += 1
        eng.accumulate_device(dX.data_ptr(), F, dy.data_ptr(), T, last=True)
        return finish()

    def step_host():
        X, y = batches[counter[0] % ring]
        counter[0] += 1
        eng.accumulate(X, y, last=True)  # pinned double-buffered staging + H2D on the copy stream (tfk_accumulate)
        return finish()

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if dp.enabled:
            dist.barrier()
            torch.cuda.synchronize()

    # Bring the GPU to its sustained clocks before anything is measured: a few hundred milliseconds of unrelated fp32
    # GEMMs on scratch buffers (no model state is touched, the loss trace still starts at the initial weights).  With a
    # short run (the driver's --steps 20 --warmup 5 is 40 ms of GPU work) the first timed steps otherwise run while
    # the clocks are still ramping up from idle: ~3 % slower than the steady state the metric is about.
    prewarm_ms = float(os.environ.get("TFK_BENCH_PREWARM_MS", "300"))
    if prewarm_ms > 0:
        import ctypes
        sa = torch.randn(1024, 2048, device="cuda"); sb = torch.randn(2048, 2048, device="cuda")
        sc = torch.empty(1024, 2048, device="cuda")
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        t_end = time.perf_counter() + prewarm_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(50):
                _lib.check(eng.lib.tfk_gemm_f32(stream, 0, ctypes.c_void_p(sa.data_ptr()), 2048, ctypes.c_void_p(sb.data_ptr()),
                                                2048, ctypes.c_void_p(sc.data_ptr()), 2048, 1024, 2048, 2048, None, 0, -1))
            torch.cuda.synchronize()
        del sa, sb, sc
    losses = [step() for _ in range(args.warmup)]
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(step())
    fence()
    elapsed = time.perf_counter() - t0
    my_ms = 1e3 * elapsed / args.steps
    # the same step fed from host memory (SURVEY 8d / the reference's feed_dict), a different micro-batch per step
    for _ in range(min(3, args.steps)):
        step_host()
    fence()
    th = time.perf_counter()
    for _ in range(args.steps):
        step_host()
    fence()
    elapsed_host = time.perf_counter() - th
    # Per-kernel HIP-event timing on the engine stream, over the same K steps repeated right after the timed
    # regions: bracketing every launch with two events costs ~10 % of wall time, so it is kept out of `value`.
    eng.profile_begin()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed_profiled = time.perf_counter() - t1
    stats = eng.profile_end()
    per_rank_ms = [my_ms]
    backend = None
    if dp.enabled:
        t_all = torch.tensor([elapsed, elapsed_host], dtype=torch.float64, device="cuda")
        gathered = [torch.zeros_like(t_all) for _ in range(world)]
        dist.all_gather(gathered, t_all)
        per_rank_ms = [1e3 * float(g[0].item()) / args.steps for g in gathered]
        elapsed = max(float(g[0].item()) for g in gathered)
        elapsed_host = max(float(g[1].item()) for g in gathered)
        backend = dist.get_backend()

    if rank == 0:
        gemms = [s for s in stats if s["name"].startswith("gemm_f32")]
        dom = max(gemms, key=lambda s: s["total_ms"])
        achieved = dom["flops"] / dom["total_ms"] / 1e9  # TFLOP/s
        all_gemm_tf = sum(s["flops"] for s in gemms) / sum(s["total_ms"] for s in gemms) / 1e9
        value = world * T * args.steps / elapsed
        peak = PEAK_FP32_MFMA_TFLOPS if args.dtype == "float32" else PEAK_BF16_MFMA_TFLOPS
        traffic, traffic_src = (None, "fp32 only") if args.dtype != "float32" else measured_traffic(dom["name"])
        out = {
            "metric": "acoustic frames/sec (train step)", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "clock_prewarm_ms": prewarm_ms,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_with_event_profiling": 1e3 * elapsed_profiled / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.dtype == "float32" else "bf16 operands, f32 accumulate / master / optimiser",
            "data": "synthetic ark/scp/alignment files (SURVEY 8d) read through the feature reader + dispenser, %d "
                    "distinct micro-batches per rank cycled (a different one every step); random-init weights "
                    "N(0,1/sqrt(d_in)), zero output layer" % ring,
            "config": {"workload": "cfg2: 6x2048 ReLU+BN DNN, 440-in (40 fbank +-5), 2000 pdf, %d frames/GPU/step, "
                                   "%s, Adam" % (T, "fp32 MFMA" if args.dtype == "float32" else "bf16 MFMA (mixed precision)"),
                       "frames_per_gpu": T, "global_frames": world * T,
                       "parallelism": "dp%d" % world, "flop_per_frame": FLOP_PER_FRAME},
            "rccl_ranks": world if backend == "nccl" else 0, "dist_backend": backend,
            # what RAN, not what was asked for: the exchange mode the reducer settled on after probing the backend and
            # the torch.distributed calls of the last timed step, in launch order
            "exchange": reducer.mode if reducer else None,
            "exchange_requested": (args.exchange or os.environ.get("TFK_DP_EXCHANGE", "sharded")) if reducer else None,
            "collectives_last_step": list(reducer.last_executed) if reducer else None,
            "collective_spans_last_step": [list(x) for x in reducer.last_launched] if reducer else None,
            "dp_host_ms_per_step": ({k: 1e3 * v / max(1, reducer.host_calls["finish_and_apply"])
                                     for k, v in reducer.host_s.items()} if reducer else None),
            "per_rank_ms_per_step": per_rank_ms,
            "roofline": {"bound": "mfma", "kernel": dom["name"], "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
                         "launches": dom["launches"], "avg_launch_us": 1e3 * dom["total_ms"] / dom["launches"],
                         "all_gemm_tflops": all_gemm_tf,
                         "step_tflops": value / world * FLOP_PER_FRAME / 1e12,
                         "step_frac": value / world * FLOP_PER_FRAME / 1e12 / peak},
            "host_fed_value": world * T * args.steps / elapsed_host,
            "host_fed_note": "same step, micro-batch handed over as HOST numpy [1024, 440] + targets through "
                             "tfk_accumulate (PCIe inclusive; never `value`)",
            "loss_first_last": [losses[0], losses[-1]],
            "loss_trace_gpu": losses[:TRACE_STEPS],
            "kernel_ms_per_step": {s["name"]: s["total_ms"] / args.steps for s in stats},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], trace_cpu = cpu_baseline(batches, hidden)
            out["loss_trace_cpu"] = trace_cpu
            n = min(len(trace_cpu), len(out["loss_trace_gpu"]))
            if n:
                out["loss_trace_max_rel_diff"] = max(abs(a - b) / max(abs(b), 1e-30)
                                                     for a, b in zip(out["loss_trace_gpu"][:n], trace_cpu[:n]))
            if not args.no_f64_trace and args.dtype == "float32":
                # the referee (same leg as the CPU baseline: oracle code, after every timed region): the float64 oracle
                # over the same weights and micro-batches -- how far each fp32 implementation is from the specified
                # arithmetic, not merely from the other one
                from oracle.loss_trace import distances, f64_loss_trace
                t_ref = time.perf_counter()
                ref = f64_loss_trace(batches, hidden, min(TRACE_STEPS, len(out["loss_trace_gpu"])), F, L, H, O)
                out["loss_trace_f64"] = ref
                out["loss_trace_f64_seconds"] = time.perf_counter() - t_ref
                gpu_rel, _ = distances(out["loss_trace_gpu"], ref)
                cpu_rel, _ = distances(trace_cpu, ref)
                out["loss_trace_f64_max_rel_diff"] = max(gpu_rel) if gpu_rel else None
                out["loss_trace_cpu_vs_f64_max_rel_diff"] = max(cpu_rel) if cpu_rel else None
                out["loss_trace_f64_rel_diff_per_step"] = {"engine": gpu_rel, "cpu_fp32": cpu_rel}
            dp.gather_parameters(eng)
            out["posterior_max_err"] = posterior_error(eng, batches[0][0][:UTT_LEN])
    eng.close()
    if dp.enabled:
        dist.destroy_process_group()
    if rank == 0:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)  # C stdio buffers still point at the diverted fd
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
