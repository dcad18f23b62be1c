"""Benchmark of the DNN training hot path on MI355X (contract: python bench.py --gpus N --steps K --warmup W).

Workload = BASELINE.json configs[1] ("cfg2"): 6x2048 ReLU + batch-norm DNN, 40-dim fbank +-5 splice = 440
inputs, 2000 pdf-ids, 1024 frames (16 utterances x 64 frames) per GPU per optimiser step, fp32 (exact-fp32 MFMA).
One "step" = one full optimiser step of the reference's Trainer.update: forward + softmax-CE + backward on the
micro-batch, gradient all-reduce when N > 1, mean -> clip -> Adam, BN moving averages, loss returned to the host.
Weak scaling: every rank owns its own 1024-frame micro-batch (one micro-batch per GPU, as the reference's
128-utterance batch in 16-utterance micro-batches maps onto 8 GPUs); `value` = all ranks' frames / max-rank time.
Inputs come from synthetic ark/scp/alignment files through the product's feature reader + batch dispenser and
are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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


def make_batch(rank, world, workdir):
    """this rank's micro-batch through the product I/O path: ark -> CMVN -> splice -> dispenser"""
    from tfkaldi_amd import synthetic
    from tfkaldi_amd.processing import batchdispenser, feature_reader, target_coder
    paths = synthetic.write_corpus(workdir, UTT_PER_GPU * world, O, feat_dim=F_RAW, utt_len=UTT_LEN)
    reader = feature_reader.FeatureReader(paths["feats_scp"], paths["cmvn_scp"], paths["utt2spk"], CONTEXT, UTT_LEN)
    coder = target_coder.AlignmentCoder(lambda x, y: x, O)
    disp = batchdispenser.AlignmentBatchDispenser(reader, coder, UTT_PER_GPU, paths["alignments"])
    for _ in range(rank + 1):
        xs, ys = disp.get_batch()
    return np.concatenate(xs, 0), np.concatenate(ys, 0).astype(np.int32)


def cpu_baseline(X, y, hidden_weights, budget_s=20.0):
    """the same optimiser step on the host cores (PyTorch CPU restatement; TensorFlow is not available)"""
    import torch
    from oracle.torch_cpu_step import TorchCpuTrainer
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 64)  # beyond ~64 threads the 1024-row GEMMs of this step stop scaling on the host
    t = TorchCpuTrainer(F, L, H, O, nonlin="relu", batch_norm=True, threads=cores)
    t.set_hidden_weights(hidden_weights)
    t.accumulate(X, y); t.apply()  # warm-up
    steps, t0 = 0, time.perf_counter()
    while True:
        t.accumulate(X, y); t.apply()
        steps += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or steps >= 50:
            break
    return {"value": steps * T / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d optimiser steps of the same cfg2 micro-batch (%d frames each) after 1 warm-up, "
                      "PyTorch-CPU fp32 restatement of CrossEnthropyTrainer.update (TensorFlow absent)" % (steps, T)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=["float32", "bfloat16"], default="float32",
                    help="float32 (default) is BASELINE cfg2's arithmetic and the only valid headline; bfloat16 runs "
                         "the same workload in the engine's mixed-precision mode (cfg3/cfg4 arithmetic) for reference")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON record: libraries that chat on stdout (RCCL prints its library
    # path from C stdio at teardown) are diverted to stderr until the record is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from tfkaldi_amd import _lib
    from tfkaldi_amd.dataparallel import BucketReducer, DataParallel, init_from_env
    from tfkaldi_amd.engine import Engine

    rank, world, local_rank = init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d (launch with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)  # (init_from_env folds local_rank onto one device under TFK_SHARE_DEVICE)
    dp = DataParallel()

    with tempfile.TemporaryDirectory(prefix="tfkaldi_bench_") as workdir:
        X, y = make_batch(rank, world, os.path.join(workdir, "rank%d" % rank))
    assert X.shape == (T, F) and X.dtype == np.float32 and y.shape == (T,)

    cfg = _lib.make_config(F, L, H, O, nonlin="relu", batch_norm=True, init_learning_rate=1e-3,
                           num_steps=args.steps + args.warmup, max_frames=T, device=local_rank,
                           compute_dtype=args.dtype)
    eng = Engine(cfg, torch_state=dp.enabled)
    rng = np.random.default_rng(7)
    hidden = [(rng.standard_normal((F if l == 0 else H, H)) / np.sqrt(F if l == 0 else H)).astype(np.float32)
              for l in range(L)]
    for l, w in enumerate(hidden):
        eng.set(_lib.WEIGHTS, l, w)

    dX = torch.from_numpy(X).cuda()
    dy = torch.from_numpy(y).cuda()
    torch.cuda.synchronize()

    if dp.enabled:
        import torch.distributed as dist
        # the product's exchange step: per-layer bucket announcements from backward, coalesced into a few large
        # asynchronous all-reduces (tfkaldi_amd/dataparallel.py)
        reducer = BucketReducer(eng, stream_ctx=lambda: torch.cuda.stream(eng.torch_stream))
        eng.set_bucket_callback(reducer.on_bucket)
        eng.set_later_microbatches(world - 1 - rank)

    def step():
        eng.accumulate_device(dX.data_ptr(), F, dy.data_ptr(), T, last=True)
        if dp.enabled:
            return reducer.finish_and_apply(eng)  # Adam per reduced span, behind the collectives still in flight
        return eng.apply()

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if dp.enabled:
            dist.barrier()
            torch.cuda.synchronize()

    losses = [step() for _ in range(args.warmup)]
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(step())
    fence()
    elapsed = time.perf_counter() - t0
    # Per-kernel HIP-event timing on the engine stream, over the same K steps repeated right after the timed
    # region: bracketing every launch with two events costs ~10 % of wall time, so it is kept out of `value`.
    eng.profile_begin()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed_profiled = time.perf_counter() - t1
    stats = eng.profile_end()
    if dp.enabled:
        t_max = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        elapsed = float(t_max.item())

    if rank == 0:
        gemms = [s for s in stats if s["name"].startswith("gemm_f32")]
        dom = max(gemms, key=lambda s: s["total_ms"])
        achieved = dom["flops"] / dom["total_ms"] / 1e9  # TFLOP/s
        all_gemm_tf = sum(s["flops"] for s in gemms) / sum(s["total_ms"] for s in gemms) / 1e9
        value = world * T * args.steps / elapsed
        peak = PEAK_FP32_MFMA_TFLOPS if args.dtype == "float32" else PEAK_BF16_MFMA_TFLOPS
        traffic, traffic_src = None, None
        tfile = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if os.path.exists(tfile) and args.dtype == "float32":  # separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command
            rec = json.load(open(tfile)).get(dom["name"])
            if rec:
                traffic = rec["bytes_per_launch"]
                traffic_src = ("profiles/r01_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), "
                               "(2*FETCH+WRITE)*1024 per MI355X_MICROARCH; L2<->fabric side, Infinity-Cache hits included")
        out = {
            "metric": "acoustic frames/sec (train step)", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_with_event_profiling": 1e3 * elapsed_profiled / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.dtype == "float32" else "bf16 operands, f32 accumulate / master / optimiser",
            "data": "synthetic ark/scp/alignment files (SURVEY 8d) read through the feature reader + dispenser; "
                    "random-init weights N(0,1/sqrt(d_in)), zero output layer",
            "config": {"workload": "cfg2: 6x2048 ReLU+BN DNN, 440-in (40 fbank +-5), 2000 pdf, %d frames/GPU/step, "
                                   "%s, Adam" % (T, "fp32 MFMA" if args.dtype == "float32" else "bf16 MFMA (mixed precision)"),
                       "frames_per_gpu": T, "global_frames": world * T,
                       "parallelism": "dp%d" % world, "flop_per_frame": FLOP_PER_FRAME},
            "roofline": {"bound": "mfma", "kernel": dom["name"], "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
                         "launches": dom["launches"], "avg_launch_us": 1e3 * dom["total_ms"] / dom["launches"],
                         "all_gemm_tflops": all_gemm_tf,
                         "step_tflops": value / world * FLOP_PER_FRAME / 1e12},
            "loss_first_last": [losses[0], losses[-1]],
            "kernel_ms_per_step": {s["name"]: s["total_ms"] / args.steps for s in stats},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(X, y, hidden)
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
