"""Benchmark of the DNN training hot path on MI355X (contract: python bench.py --gpus N --steps K --warmup W).

Workload = BASELINE.json configs[1] ("cfg2"): 6x2048 ReLU + batch-norm DNN, 40-dim fbank +-5 splice = 440
inputs, 2000 pdf-ids, 1024 frames (16 utterances x 64 frames) per GPU per optimiser step, fp32 -- since round 5 EMULATED on
the bf16 matrix pipe (operands split exactly into three bf16 planes, six plane products, fp32 accumulate: the product's
`compute_dtype = float32`; the judge's round-4 ruling and its conditions: `dtype` / `config.workload` say so, `roofline.peak` is
2500 / 6 TF, and the exact-fp32 MFMA figure of rounds 1-4 stays in the line as the `exact_fp32` sub-record;
`--dtype float32_mfma` makes that arithmetic the headline again).
One "step" = one full optimiser step of the reference's Trainer.update: forward + softmax-CE + backward on the
micro-batch, gradient exchange when N > 1, mean -> clip -> Adam, BN moving averages, loss returned to the host.
Weak scaling: every rank owns its own 1024-frame micro-batch (one micro-batch per GPU, as the reference's
128-utterance batch in 16-utterance micro-batches maps onto 8 GPUs); `value` = all ranks' frames / max-rank time.
Inputs come from synthetic ark/scp/alignment files through the product's feature reader + batch dispenser; a ring
of DISTINCT micro-batches (a different one every step) is resident in HBM before the timed region.

`--gpus N` with N > 1 from a bare shell launches N ranks itself (torch.distributed.run, one rank per GPU over RCCL);
under torchrun / the driver's own launcher (WORLD_SIZE set) it joins that group.  Prints ONE JSON line on rank 0.

`--config cfg3|cfg4` (or TFK_BENCH_CONFIG) runs BASELINE configs[2] / [3] as one GPU of their 8-GPU runs sees them
(bf16 MFMA, 4000 / 8000 pdfs, 1024 / 2048 frames per GPU); `--gpus 8 --config cfg3` is configs[2] itself.

Besides the contract's fields the line carries: `api_fed_value` (frames/s of Nnet.train itself over ark
files: dispenser, host, PCIe, engine -- tools/nnet_train_bench.py), `decode` (log-likelihood passes host to host + the
forward contractions' roofline), `roofline` (dominant kernel, live HIP-event timing),
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

F_RAW, CONTEXT, UTT_PER_GPU = 40, 5, 16
F = F_RAW * (2 * CONTEXT + 1)
PEAK_FP32_MFMA_TFLOPS = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0           # v_mfma_f32_32x32x16_bf16, dense
TRACE_STEPS = 20                         # per-step average_loss of the first 20 steps
CPU_WARMUP = 10                          # warm-up steps kept out of the cpu_baseline rate
MAX_RING = 16                            # distinct micro-batches resident per rank


class Workload(object):
    """One BASELINE.json configuration as ONE GPU sees it (weak scaling: the per-GPU micro-batch is fixed).
    cfg2 = configs[1], the configuration the metric is quoted on and the default; cfg3 / cfg4 = configs[2] / [3], defined
    as 8-GPU data-parallel runs (global batch 8192 / 16384 frames = 1024 / 2048 per GPU)."""

    # "float32" = the product's default fp32 arithmetic (emulated on the bf16 pipe, tfkaldi_amd/_lib.py: DTYPES);
    # "float32_mfma" = the exact fp32 matrix instructions; "bfloat16" = mixed precision
    PEAK = {"float32": PEAK_BF16_MFMA_TFLOPS / 6.0, "float32_mfma": PEAK_FP32_MFMA_TFLOPS, "bfloat16": PEAK_BF16_MFMA_TFLOPS}
    DTYPE_TEXT = {"float32": "f32 emulated (3xbf16 planes, 6 products, f32 accumulate)", "float32_mfma": "f32 (exact fp32 MFMA)",
                  "bfloat16": "bf16 operands, f32 accumulate / master / optimiser"}
    ARITH_TEXT = {"float32": "fp32 emulated on the bf16 MFMA pipe (operands split exactly into 3 bf16 planes, 6 plane products, "
                             "fp32 accumulate)", "float32_mfma": "exact fp32 MFMA", "bfloat16": "bf16 MFMA (mixed precision)"}
    TABLE = {  # name: (hidden layers, units, pdfs, frames per GPU per step, dropout keep, arithmetic, description)
        "cfg2": (6, 2048, 2000, 1024, 1.0, "float32", "cfg2: 6x2048 ReLU+BN DNN, 440-in (40 fbank +-5), 2000 pdf"),
        "cfg3": (6, 2048, 4000, 1024, 1.0, "bfloat16", "cfg3: 6x2048 ReLU+BN DNN, 440-in, 4000 pdf (lda_mllt), "
                                                        "global batch 8192 over 8 GPUs"),
        "cfg4": (8, 4096, 8000, 2048, 0.5, "bfloat16", "cfg4: 8x4096 ReLU+BN+Dropout(0.5) DNN, 440-in, 8000 pdf "
                                                        "(CGN scale), global batch 16384 over 8 GPUs"),
    }

    def __init__(self, name, dtype=None):
        self.name = name
        self.L, self.H, self.O, self.T, self.keep, default_dtype, self.text = self.TABLE[name]
        self.dtype = {"float32x3": "float32"}.get(dtype or default_dtype, dtype or default_dtype)
        if self.dtype == "float32" and os.environ.get("TFK_F32_ARITHMETIC") == "mfma":
            self.dtype = "float32_mfma"  # (the process-wide policy switch of _lib.resolve_dtype: say what will really run)
        self.utt_len = self.T // UTT_PER_GPU
        self.macs = F * self.H + (self.L - 1) * self.H * self.H + self.H * self.O
        # fwd 2M, dW 2M, dA 2M minus the unneeded layer-0 dA
        self.flop_per_frame = 6 * self.macs - 2 * F * self.H
        # (emulated fp32: six bf16 MFMAs per fp32 product -- the ceiling of the emulation in fp32-equivalent flops)
        self.peak = self.PEAK[self.dtype]


def self_launch(args):
    """`python bench.py --gpus N` from a bare shell: re-execute under torch.distributed.run, one rank per GPU."""
    import torch
    # tests only: several ranks on one GPU (over gloo, or over real RCCL with every rank claiming its own host)
    share = os.environ.get("TFK_SHARE_DEVICE") == "1" or os.environ.get("TFK_FAKE_NODES") == "1"
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


def make_batches(w, rank, world, count, workdir):
    """`count` micro-batches of this rank through the product I/O path (ark -> CMVN -> splice -> dispenser): step i
    of the job is the reference's batch i (world * 16 utterances in dispenser order), of which rank r takes the
    r-th 16-utterance micro-batch."""
    from tfkaldi_amd import synthetic
    from tfkaldi_amd.processing import batchdispenser, feature_reader, target_coder
    paths = synthetic.write_corpus(workdir, UTT_PER_GPU * world * count, w.O, feat_dim=F_RAW, utt_len=w.utt_len)
    reader = feature_reader.FeatureReader(paths["feats_scp"], paths["cmvn_scp"], paths["utt2spk"], CONTEXT, w.utt_len)
    coder = target_coder.AlignmentCoder(lambda x, y: x, w.O)
    disp = batchdispenser.AlignmentBatchDispenser(reader, coder, UTT_PER_GPU, paths["alignments"])
    out = []
    for g in range(world * count):
        xs, ys = disp.get_batch()
        if g % world == rank:
            out.append((np.ascontiguousarray(np.concatenate(xs, 0)), np.concatenate(ys, 0).astype(np.int32)))
    return out


def cpu_baseline(w, batches, hidden_weights, budget_s=20.0):
    """the same optimiser steps on the host cores (PyTorch-CPU restatement; TensorFlow is not installable here)"""
    import torch
    from oracle.torch_cpu_step import TorchCpuTrainer
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 64)  # beyond ~64 threads the 1024-row GEMMs of this step stop scaling on the host
    t = TorchCpuTrainer(F, w.L, w.H, w.O, nonlin="relu", batch_norm=True, threads=cores)
    T = w.T
    t.set_hidden_weights(hidden_weights)
    # the first CPU_WARMUP steps (thread pool start, allocator, first-touch pages) belong to the loss trace but not to the rate
    trace, steps, t0, t_warm = [], 0, time.perf_counter(), None
    while True:
        X, y = batches[steps % len(batches)]
        t.accumulate(X, y)
        trace.append(t.apply())
        steps += 1
        now = time.perf_counter()
        if steps == CPU_WARMUP:
            t_warm = now
        dt = now - t0
        if (dt > budget_s and steps >= TRACE_STEPS) or steps >= 60 or dt > 3 * budget_s:
            break
    timed = steps - CPU_WARMUP if t_warm is not None and steps > CPU_WARMUP else steps
    span = now - t_warm if timed != steps else dt
    return {"value": timed * T / span, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d optimiser steps from the same initial weights over the same micro-batch sequence (%d frames "
                      "each), the first %d untimed, PyTorch-CPU fp32 restatement of CrossEnthropyTrainer.update (a stand-in: "
                      "TensorFlow, the reference's CPU path, is absent)%s" % (
                          steps, T, steps - timed, "" if w.keep >= 1 else "; the stand-in has no dropout layer (the masks cost it "
                          "nothing measurable)")}, trace[:TRACE_STEPS]


def posterior_error(w, eng, X):
    """max |engine posterior - float64 oracle posterior| on one utterance, the oracle holding the engine's parameters"""
    from oracle.dnn_oracle import OracleDNN
    from tfkaldi_amd import _lib
    L = w.L
    orc = OracleDNN(F, L, w.H, w.O, nonlin="relu", batch_norm=True)
    for l in range(L + 1):
        orc.W[l] = eng.get(_lib.WEIGHTS, l).astype(np.float64)
        orc.b[l] = eng.get(_lib.BIASES, l).astype(np.float64)
    for l in range(L):
        orc.beta[l] = eng.get(_lib.BN_BETA, l).astype(np.float64)
        orc.mov_mean[l] = eng.get(_lib.BN_MOVING_MEAN, l).astype(np.float64)
        orc.mov_var[l] = eng.get(_lib.BN_MOVING_VAR, l).astype(np.float64)
    return float(np.abs(eng.posteriors(X).astype(np.float64) - orc.posteriors(X)).max())


def measured_traffic(w, kernel_label_):
    """(bytes per launch of the dominant kernel, the record's per-step summary, where it comes from): the committed PMC summary of
    THIS configuration and arithmetic (tools/hbm_counters.sh + tools/hbm_traffic.py), valid only for the kernel sources it was
    measured on."""
    from tfkaldi_amd.build import csrc_hash
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(path):
        return None, None, "no profiles/hbm_traffic.json"
    rec = json.load(open(path)).get("%s/%s" % (w.name, w.dtype))
    if rec is None:
        return None, None, "no record for %s/%s in profiles/hbm_traffic.json" % (w.name, w.dtype)
    meta = rec.get("_meta", {})
    if meta.get("csrc_sha16") != csrc_hash():
        return None, None, "profiles/hbm_traffic.json[%s/%s] is stale (measured on csrc %s, this build is %s): dropped" % (
            w.name, w.dtype, meta.get("csrc_sha16"), csrc_hash())
    src = ("profiles/hbm_traffic.json[%s/%s] (csrc %s, %s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this "
           "command, (2*FETCH+WRITE)*1024 per MI355X_MICROARCH.md; L2<->fabric side, Infinity-Cache hits included"
           % (w.name, w.dtype, meta.get("csrc_sha16"), meta.get("measured", "?")))
    if kernel_label_ not in rec:
        return None, meta, src + "; dominant kernel %s not in the record" % kernel_label_
    return rec[kernel_label_]["bytes_per_launch"], meta, src


def rocprof_reference(w):
    """the dominant kernel's average launch by `rocprofv3 --kernel-trace --stats` of this command as committed under profiles/
    (tools/kernel_stats_txt.py writes profiles/kernel_stats_ref.json), quoted BESIDE this run's own HIP-event figure -- the two
    must agree (round-5 verdict: state both in the line).  None / a reason when there is no record for this build's GEMM sources"""
    from tfkaldi_amd.build import csrc_hash
    path = os.path.join(ROOT, "profiles", "kernel_stats_ref.json")
    if not os.path.exists(path):
        return None
    rec = json.load(open(path)).get("%s/%s" % (w.name, w.dtype))
    if rec is None:
        return None
    if rec.get("csrc_sha16") != csrc_hash():
        return {"stale": "measured on csrc %s, this build is %s" % (rec.get("csrc_sha16"), csrc_hash()), "source": rec.get("source")}
    return {"avg_launch_us": rec["avg_launch_us"], "kernel": rec["kernel"], "calls": rec["calls"], "source": "profiles/" + rec["source"]}


def other_arithmetic_leg(w, other, batches, hidden, steps, warmup, device, ref_trace):
    """The same workload in the OTHER fp32 arithmetic (w.dtype float32 = emulated on the bf16 pipe -> the exact fp32 matrix
    instructions, reported as `exact_fp32`; and vice versa, `emulated_fp32`).  Reported beside the headline, never as it.  Same
    weights, same micro-batch sequence; its loss trace is held against the same float64 referee."""
    import torch
    from tfkaldi_amd import _lib
    from tfkaldi_amd.engine import Engine
    cfg = _lib.make_config(F, w.L, w.H, w.O, nonlin="relu", batch_norm=True, keep_prob=w.keep, init_learning_rate=1e-3,
                           num_steps=3 * (steps + warmup), max_frames=w.T, device=device,
                           compute_dtype={"float32": "float32x3"}.get(other, other))
    eng = Engine(cfg)
    for l, weights in enumerate(hidden):
        eng.set(_lib.WEIGHTS, l, weights)
    dev = [(torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()) for X, y in batches]
    torch.cuda.synchronize()
    n = [0]

    def step():
        dX, dy = dev[n[0] % len(dev)]
        n[0] += 1
        eng.accumulate_device(dX.data_ptr(), F, dy.data_ptr(), w.T, last=True)
        return eng.apply()

    losses = [step() for _ in range(warmup)]
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(step())
    eng.synchronize()
    dt = (time.perf_counter() - t0) / steps
    eng.profile_begin()
    for _ in range(min(steps, 20)):
        step()
    eng.synchronize()
    stats = eng.profile_end()
    eng.close()
    ow = Workload(w.name, other)
    gemms = [s for s in stats if s["name"].startswith("gemm_")]
    dom = max(gemms, key=lambda s: s["total_ms"])
    tf = dom["flops"] / dom["total_ms"] / 1e9
    out = {"value": w.T / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt, "dtype": Workload.DTYPE_TEXT[other],
           "arithmetic": Workload.ARITH_TEXT[other] + "; parameters / statistics / loss / gradient sums / Adam in fp32",
           "roofline": {"bound": "mfma", "kernel": kernel_label(ow, dom["name"]), "achieved": tf, "peak": ow.peak, "unit": "TFLOP/s",
                        "frac": tf / ow.peak, "avg_launch_us": 1e3 * dom["total_ms"] / dom["launches"],
                        "all_gemm_tflops": sum(s["flops"] for s in gemms) / sum(s["total_ms"] for s in gemms) / 1e9,
                        "step_tflops": w.T / dt * w.flop_per_frame / 1e12,
                        "step_frac": w.T / dt * w.flop_per_frame / 1e12 / ow.peak},
           "loss_trace": losses[:TRACE_STEPS],
           "kernel_ms_per_step": {kernel_label(ow, s["name"]): s["total_ms"] / min(steps, 20) for s in stats}}
    if ref_trace:
        from oracle.loss_trace import distances
        rel, _ = distances(losses, ref_trace)
        out["loss_trace_f64_max_rel_diff"] = max(rel) if rel else None
        out["loss_trace_f64_rel_diff_per_step"] = rel
    return out


def decode_leg(w, eng, batches):
    """The decode half of the path (reference neuralNetworks/decoder.py:49-71, nnet.py:270-286): evaluation-mode forward
    + softmax / prior + log -> log-likelihoods on the HOST, for one batched pass of 8 micro-batches and for
    utterance-sized passes.  Host frames in, host log-likelihoods out (PCIe both ways inside the clock); the roofline of
    the forward contractions (2 * M flop per frame) from the engine's HIP-event profile of the batched pass."""
    X = np.ascontiguousarray(np.concatenate([b[0] for b in batches[:8]], 0))
    eng.set_prior(np.full(w.O, 1.0 / w.O, dtype=np.float32))
    out = {"unit": "frames/s", "flop_per_frame": 2 * w.macs, "passes": {}}

    def rate(frames, reps):
        """frames / MEDIAN pass time (a pass of a new size grows the engine's buffers and the pinned result once)"""
        Xp = X[:frames]
        eng.posteriors(Xp, log_div_prior=True)
        eng.posteriors(Xp, log_div_prior=True)
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            eng.posteriors(Xp, log_div_prior=True)
            times.append(time.perf_counter() - t0)
        return frames / sorted(times)[len(times) // 2]

    for frames, reps in ((300, 20), (1000, 10), (2000, 8), (X.shape[0], 5)):
        if frames <= X.shape[0]:
            out["passes"]["%d frames" % frames] = rate(frames, reps)
    out["value"] = out["passes"]["%d frames" % X.shape[0]]
    out["batched_pass_frames"] = int(X.shape[0])
    eng.profile_begin()
    for _ in range(3):
        eng.posteriors(X, log_div_prior=True)
    eng.synchronize()
    stats = eng.profile_end()
    gemms = [s for s in stats if s["name"].startswith("gemm_")]
    if gemms:
        dom = max(gemms, key=lambda s: s["total_ms"])
        tf = dom["flops"] / dom["total_ms"] / 1e9
        device_ms = sum(s["total_ms"] for s in stats) / 3
        out["roofline"] = {"bound": "mfma", "kernel": kernel_label(w, dom["name"]), "achieved": tf, "peak": w.peak,
                           "unit": "TFLOP/s", "frac": tf / w.peak,
                           "avg_launch_us": 1e3 * dom["total_ms"] / dom["launches"],
                           "device_ms_per_batched_pass": device_ms,
                           "device_only_frames_per_s": X.shape[0] / (device_ms * 1e-3),
                           "device_only_step_frac": X.shape[0] / (device_ms * 1e-3) * 2 * w.macs / 1e12 / w.peak}
        out["note"] = ("`value` and `passes` are host-to-host (frames over PCIe in, %d B of log-likelihoods per frame out); "
                       "device_only_* is the same batched pass from the engine's kernel times alone" % (4 * w.O))
    return out


def eval_leg(w, eng, batches):
    """Validation (reference neuralNetworks/trainer.py:356-441, nnet.py:168-207): the loss of 8 held-out micro-batches in
    evaluation mode, host frames in, as ONE stacked pass (tfk_eval_accumulate_stacked: rows are independent in evaluation mode)
    and as the reference's one run per micro-batch.  Median of 5; frames/s and the forward contractions' share of the matrix peak."""
    mbs = batches[:8]
    Xs = np.ascontiguousarray(np.concatenate([b[0] for b in mbs], 0))
    ys = np.concatenate([b[1] for b in mbs], 0)
    rows = [len(b[1]) for b in mbs]

    def stacked():
        eng.eval_accumulate_stacked(Xs, ys, rows)
        return eng.eval_finish()

    def one_by_one():
        for X, y in mbs:
            eng.eval_accumulate(X, y)
        return eng.eval_finish()

    def med(fn):
        fn(); fn()
        times = []
        for _ in range(5):
            t0 = time.perf_counter()
            loss = fn()
            times.append(time.perf_counter() - t0)
        return sorted(times)[2], loss
    t_st, l_st = med(stacked)
    t_seq, l_seq = med(one_by_one)
    frames = int(Xs.shape[0])
    return {"unit": "frames/s", "frames": frames, "microbatches": len(mbs), "value": frames / t_st,
            "one_pass_per_microbatch": frames / t_seq, "stacked_over_sequential": t_seq / t_st,
            "loss_stacked": l_st, "loss_sequential": l_seq, "flop_per_frame": 2 * w.macs,
            "step_frac_of_peak": frames / t_st * 2 * w.macs / 1e12 / w.peak,
            "note": "host numpy in (PCIe inside the clock), loss out; forward only"}


class PowerSampler(object):
    """socket power (W) and shader clock (MHz) of this rank's GPU, sampled from a host thread while a leg runs.  The amdgpu hwmon
    files when this user can read them (a read costs microseconds: one sample every 50 ms), else `rocm-smi --showpower
    --showclocks` as tools/x3_power_sample.sh uses it (a Python process per sample: ~3 samples a second)."""

    def __init__(self, device_index):
        import glob
        import threading
        self.samples, self.source, self._stop = [], None, threading.Event()
        self._power = self._clock = None
        self.bdf = self._pci_bus_id(device_index)
        self.smi_index = None
        # the box may expose more cards in sysfs than this process can use: the hwmon directory is picked by PCI address
        for hw in sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*")):
            if self.bdf is None or not os.path.realpath(os.path.dirname(os.path.dirname(hw))).lower().endswith(self.bdf):
                continue
            for name in ("power1_input", "power1_average"):
                f = os.path.join(hw, name)
                try:
                    if float(open(f).read()) > 0:
                        self._power = f
                        break
                except (OSError, ValueError):
                    pass
            try:
                float(open(os.path.join(hw, "freq1_input")).read())
                self._clock = os.path.join(hw, "freq1_input")
            except (OSError, ValueError):
                pass
            break
        if not (self._power and self._clock):
            self._power = self._clock = None
            self.smi_index = self._smi_index()
        self.source = ("amdgpu hwmon of %s (power1, freq1_input)" % self.bdf if self._power
                       else "rocm-smi -d %s --showpower --showclocks (PCI %s)" % (self.smi_index, self.bdf))
        self._thread = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _pci_bus_id(device_index):
        """'dddd:bb:dd.f' of HIP device `device_index` (hipDeviceGetPCIBusId), lower case; None when it cannot be asked"""
        try:
            import ctypes
            import torch
            hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) != 0:
                return None
            return buf.value.decode().strip().lower() or None
        except Exception:  # noqa: BLE001
            return None

    def _smi_index(self):
        """rocm-smi's index of the GPU at self.bdf (its numbering covers every card the box shows, not the visible ones)"""
        import re
        try:
            out = subprocess.run(["rocm-smi", "--showbus"], capture_output=True, text=True, timeout=20).stdout
        except Exception:  # noqa: BLE001
            return 0
        for line in out.splitlines():
            m = re.search(r"GPU\[(\d+)\].*PCI Bus:\s*(\S+)", line)
            if m and self.bdf and m.group(2).lower() == self.bdf:
                return int(m.group(1))
        return 0

    def _once(self):
        if self._power and self._clock:
            return float(open(self._power).read()) * 1e-6, float(open(self._clock).read()) * 1e-6
        import re
        out = subprocess.run(["rocm-smi", "-d", str(self.smi_index or 0), "--showpower", "--showclocks"], capture_output=True,
                             text=True, timeout=10).stdout
        pw = re.search(r"Package Power \(W\):\s*([0-9.]+)", out)
        ck = re.search(r"sclk clock level:.*\((\d+)Mhz\)", out)
        return (float(pw.group(1)) if pw else None), (float(ck.group(1)) if ck else None)

    def _run(self):
        while not self._stop.is_set():
            try:
                pw, ck = self._once()
                self.samples.append((time.perf_counter(), pw, ck))
            except Exception:  # noqa: BLE001  (a sampler must never take the bench line down)
                pass
            self._stop.wait(0.05 if self._power else 0.1)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(timeout=15)

    def summary(self, t_from):
        pw = [p for t, p, _ in self.samples if t >= t_from and p]
        ck = [c for t, _, c in self.samples if t >= t_from and c]
        return {"sampler": self.source, "samples": len(pw), "pci_bus_id": self.bdf,
                "socket_power_w_mean": sum(pw) / len(pw) if pw else None, "socket_power_w_max": max(pw) if pw else None,
                "shader_clock_mhz_mean": sum(ck) / len(ck) if ck else None, "shader_clock_mhz_min": min(ck) if ck else None}


def clock_prewarm(eng, dtype, ms):
    """Bring the GPU to its sustained clocks before anything is measured, ON THE PIPE UNDER TEST (round 5 ran fp32 MFMA GEMMs in
    front of a bf16-pipe workload): `ms` milliseconds of the arithmetic's own contraction -- the three-plane emulated GEMM, the
    bf16 GEMM, or the exact fp32 one -- on scratch operands with random significands (no model state is touched; the loss trace
    still starts at the initial weights).  With a short run (the driver's --steps 20 --warmup 5 is 40 ms of GPU work) the first
    timed steps otherwise run while clock and power management are still settling."""
    import ctypes
    import torch
    from tfkaldi_amd import _lib, x3
    M, N, K = 1024, 2048, 2048
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(K, N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if dtype == "float32":
        Ap, lda = x3.split(eng.lib, A)
        Bp, ldb = x3.split(eng.lib, B)
        call = lambda: eng.lib.tfk_gemm_bf16x3(stream, 0, ctypes.c_void_p(Ap.data_ptr()), lda, ctypes.c_void_p(Bp.data_ptr()), ldb,  # noqa: E731
                                               ctypes.c_void_p(C.data_ptr()), N, M, N, K, None, 0)
    elif dtype == "bfloat16":
        Ab, Bb = A.bfloat16(), B.bfloat16()
        call = lambda: eng.lib.tfk_gemm_bf16(stream, 0, ctypes.c_void_p(Ab.data_ptr()), K, ctypes.c_void_p(Bb.data_ptr()), N,  # noqa: E731
                                             ctypes.c_void_p(C.data_ptr()), N, M, N, K, None, 0)
    else:
        call = lambda: eng.lib.tfk_gemm_f32(stream, 0, ctypes.c_void_p(A.data_ptr()), K, ctypes.c_void_p(B.data_ptr()), N,  # noqa: E731
                                            ctypes.c_void_p(C.data_ptr()), N, M, N, K, None, 0, -1)
    t_end = time.perf_counter() + ms * 1e-3
    while time.perf_counter() < t_end:
        for _ in range(50):
            _lib.check(call())
        torch.cuda.synchronize()


def kernel_label(w, family):
    """the engine names its kernel families after the fp32 kernels; say which arithmetic actually ran"""
    return family.replace("gemm_f32", {"float32": "gemm_bf16x3", "float32_mfma": "gemm_f32", "bfloat16": "gemm_bf16"}[w.dtype])


def api_fed_leg(w, world, steps):
    """`Nnet.train` itself (tools/nnet_train_bench.py): the reference's entry point over ark files, dispenser included.
    COLLECTIVE under data parallelism: every rank calls it."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from nnet_train_bench import measure
    out = {}
    # the same work per step as `value`: one 16-utterance micro-batch per GPU per optimiser step
    r = measure(w.name, UTT_PER_GPU * world, UTT_PER_GPU, steps + 8, 8, packed=True, utt_len=w.utt_len,
                compute_dtype=w.dtype)
    out["api_fed_value"], out["api_fed_ms_per_step"] = r["value"], r["ms_per_step"]
    if world == 1:
        # the reference's recipe (config_AURORA4.cfg:134-141): 128 utterances per step in micro-batches of 16, fed the
        # packed way and the reference's way (trainer.update(*dispenser.get_batch()))
        for key, packed in (("api_fed_value_recipe", True), ("api_fed_value_recipe_list_feed", False)):
            out[key] = measure(w.name, 8 * UTT_PER_GPU, UTT_PER_GPU, 20 + 4, 4, packed=packed, utt_len=w.utt_len,
                               compute_dtype=w.dtype)["value"]
        # ... and with the recipe's validation inside the clock: 2 held-out batches evaluated every 10 steps (nnet.py:168-207)
        out["api_fed_value_recipe_with_validation"] = measure(
            w.name, 8 * UTT_PER_GPU, UTT_PER_GPU, 30 + 4, 4, packed=True, utt_len=w.utt_len, compute_dtype=w.dtype,
            valid_batches=2, valid_frequency=10)["value"]
        out["validation_cost"] = 1.0 - out["api_fed_value_recipe_with_validation"] / out["api_fed_value_recipe"]
    out["api_fed_note"] = ("frames/s of Nnet.train (reference nnet.py:80-244) on a synthetic ark corpus: ark reads, batch "
                           "dispenser, micro-batch construction, PCIe, engine, printed loss line -- everything between two "
                           "optimiser steps.  api_fed_value: %d utterances x %d frames per GPU per step (the work of "
                           "`value`); *_recipe: 128 utterances per step in 8 micro-batches" % (UTT_PER_GPU, w.utt_len))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f64-trace", action="store_true",
                    help="skip the float64-oracle loss trace (about a minute of host numpy behind the CPU baseline)")
    ap.add_argument("--config", choices=sorted(Workload.TABLE), default=os.environ.get("TFK_BENCH_CONFIG", "cfg2"),
                    help="BASELINE.json configuration as one GPU sees it: cfg2 (default; configs[1], the one the metric is "
                         "quoted on, fp32), cfg3 / cfg4 (configs[2] / [3]: the 8-GPU bf16 runs, --gpus 8).  Also "
                         "TFK_BENCH_CONFIG")
    ap.add_argument("--dtype", choices=["float32", "float32_mfma", "bfloat16", "float32x3"], default=None,
                    help="arithmetic of the GEMMs; default = the configuration's own (cfg2 float32, cfg3 / cfg4 bfloat16).  "
                         "float32 = the product's fp32: emulated on the bf16 pipe (three bf16 planes per operand, "
                         "tfkaldi_amd/_lib.py: DTYPES; float32x3 names it explicitly); float32_mfma = the exact fp32 matrix "
                         "instructions (the headline of rounds 1-4)")
    ap.add_argument("--no-other-arithmetic", "--no-emulated", dest="no_other", action="store_true",
                    help="skip the sub-record of the other fp32 arithmetic (`exact_fp32` beside an emulated headline, "
                         "`emulated_fp32` beside an exact one; fp32 runs on one GPU only)")
    ap.add_argument("--no-eval", action="store_true", help="skip the validation leg (`eval`, N = 1 only)")
    ap.add_argument("--no-sustained", action="store_true",
                    help="skip the sustained leg (>= 3 s of the same step back to back with socket power and shader clock sampled; "
                         "also TFK_BENCH_SUSTAIN_S=0)")
    ap.add_argument("--no-decode", action="store_true", help="skip the decode leg (N = 1 only)")
    ap.add_argument("--no-api-fed", action="store_true",
                    help="skip the Nnet.train leg (api_fed_value; also TFK_BENCH_API_FED=0)")
    ap.add_argument("--exchange", choices=["sharded", "allreduce"], default=None,
                    help="N > 1: reduce-scatter + sharded Adam + all-gather (default) or all-reduce + full Adam")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    w = Workload(args.config, args.dtype)
    args.dtype = w.dtype
    L, H, O, T = w.L, w.H, w.O, w.T

    # stdout carries exactly ONE line, the JSON record: libraries that chat on stdout (RCCL prints its library
    # path from C stdio at teardown) are diverted to stderr until the record is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from tfkaldi_amd import _lib
    from tfkaldi_amd.dataparallel import DEFAULT_BUCKET_MB, DataParallel, init_from_env
    from tfkaldi_amd.engine import Engine

    rank, world, local_rank = init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)  # (init_from_env folds local_rank onto one device under TFK_SHARE_DEVICE)
    dp = DataParallel()

    total_steps = args.steps + args.warmup
    ring = max(1, min(total_steps, MAX_RING))
    with tempfile.TemporaryDirectory(prefix="tfkaldi_bench_") as workdir:
        batches = make_batches(w, rank, world, ring, os.path.join(workdir, "rank%d" % rank))
    assert all(X.shape == (T, F) and X.dtype == np.float32 and y.shape == (T,) for X, y in batches)

    cfg = _lib.make_config(F, L, H, O, nonlin="relu", batch_norm=True, keep_prob=w.keep, init_learning_rate=1e-3,
                           num_steps=3 * total_steps, max_frames=T, device=local_rank,
                           compute_dtype={"float32": "float32x3"}.get(args.dtype, args.dtype))
    eng = Engine(cfg, torch_state=dp.enabled)
    rng = np.random.default_rng(7)
    hidden = [(rng.standard_normal((F if l == 0 else H, H)) / np.sqrt(F if l == 0 else H)).astype(np.float32)
              for l in range(L)]
    for l, weights in enumerate(hidden):
        eng.set(_lib.WEIGHTS, l, weights)

    dev = [(torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()) for X, y in batches]
    torch.cuda.synchronize()

    reducer = None
    if dp.enabled:
        import torch.distributed as dist
        # the product's exchange step (tfkaldi_amd/dataparallel.py): per-layer bucket announcements from backward,
        # coalesced into a few large asynchronous collectives launched while backward is still being enqueued
        dp.mode = args.exchange
        # on RCCL: NativeExchange, the collectives launched by the library itself (csrc/exchange.hip); on other backends
        # (the single-GPU dry runs over gloo) BucketReducer over torch.distributed -- same protocol
        reducer = dp.reducer(eng)  # (collective: RCCL bootstrap / probes what the backend can do)
        reducer.begin_step(eng)
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

    # Bring the GPU to its sustained clocks before anything is measured, on the pipe under test (clock_prewarm)
    prewarm_ms = float(os.environ.get("TFK_BENCH_PREWARM_MS", "300"))
    if prewarm_ms > 0:
        clock_prewarm(eng, args.dtype, prewarm_ms)
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

    # From here on everything is DIAGNOSTICS (sustained leg, exchange A/B, per-phase times, the Nnet.train leg): the contract's
    # numbers are known.  With more than one rank those legs run collectives this build could never rehearse on hardware, so a
    # watchdog guards the line: if they are not through within the budget, rank 0 prints the contract's fields with
    # `incomplete` set and every rank leaves (each rank runs its own timer: a rank stuck in a collective cannot be asked).
    watchdog, final = None, {}
    if dp.enabled:
        import threading
        budget_s = float(os.environ.get("TFK_BENCH_DIAG_BUDGET_S", "300"))
        core_ms = 1e3 * elapsed / args.steps

        def bail_out():
            if rank == 0:
                line = {"metric": "acoustic frames/sec (train step)", "value": world * T * args.steps / elapsed, "unit": "frames/s",
                        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": core_ms,
                        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": Workload.DTYPE_TEXT[args.dtype],
                        "data": "synthetic ark/scp/alignment files (SURVEY 8d) read through the feature reader + dispenser",
                        "config": {"workload": "%s, %d frames/GPU/step, %s, Adam" % (w.text, T, Workload.ARITH_TEXT[args.dtype]),
                                   "name": w.name, "frames_per_gpu": T, "global_frames": world * T, "parallelism": "dp%d" % world,
                                   "flop_per_frame": w.flop_per_frame},
                        "per_rank_ms_per_step": per_rank_ms, "dist_backend": backend,
                        "exchange": reducer.mode if reducer else None,
                        }
                line = dict(final.get("line") or line)  # (everything, when only the teardown is stuck)
                line["incomplete"] = ("the diagnostic legs behind the timed region did not finish within %.0f s "
                                      "(TFK_BENCH_DIAG_BUDGET_S); the contract's fields were measured before them" % budget_s)
                os.write(json_fd, (json.dumps(line) + "\n").encode())
            os._exit(0)

        watchdog = threading.Timer(budget_s, bail_out)
        watchdog.daemon = True
        watchdog.start()

    def timed_steps(n):
        """n steps between two fences; seconds by the slowest rank's clock"""
        fence()
        t = time.perf_counter()
        for _ in range(n):
            step()
        fence()
        dt = time.perf_counter() - t
        if dp.enabled:
            v = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
            dt = float(v.item())
        return dt

    # ---- sustained leg (after every region a rate or a roofline is quoted from): the same step() back to back for >= 3 s, the
    # rate of its last 2 s, socket power and shader clock sampled meanwhile.  A training run lasts hours (reference nnet.py:153)
    # and this step is power-bound: `value` comes from a window of tens of milliseconds, this says what it becomes.
    sustained = None
    sustain_s = float(os.environ.get("TFK_BENCH_SUSTAIN_S", "3.2"))
    if sustain_s > 0 and not args.no_sustained:
        n_sus = max(int(os.environ.get("TFK_BENCH_SUSTAIN_MIN_STEPS", "50")), int(sustain_s / (elapsed / args.steps)) + 1)  # (`elapsed` is the slowest rank's: the same count everywhere)
        marks = []
        fence()
        with PowerSampler(local_rank) as sampler:
            t_s = time.perf_counter()
            for i in range(n_sus):
                step()
                if i % 25 == 24:
                    marks.append((time.perf_counter(), i + 1))
            fence()
            t_e = time.perf_counter()
        marks.append((t_e, n_sus))
        cut = next(((t, k) for t, k in marks if t >= t_e - 2.0), marks[0])  # first mark inside the last 2 s
        tail_steps, tail_s = n_sus - cut[1], t_e - cut[0]
        if tail_steps <= 0:
            cut, tail_steps, tail_s = (t_s, 0), n_sus, t_e - t_s
        sus = {"seconds": t_e - t_s, "steps": n_sus, "last_window_s": tail_s, "last_window_steps": tail_steps,
               "ms_per_step_last_window": 1e3 * tail_s / tail_steps, "ms_per_step_whole_leg": 1e3 * (t_e - t_s) / n_sus}
        sus.update(sampler.summary(cut[0]))
        if dp.enabled:  # the slowest rank's window counts
            v = torch.tensor([sus["ms_per_step_last_window"]], dtype=torch.float64, device="cuda")
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
            sus["ms_per_step_last_window"] = float(v.item())
        # the dominant kernel while the chip is in that state: 20 event-bracketed steps right behind the leg
        eng.profile_begin()
        for _ in range(20):
            step()
        fence()
        sus_stats = eng.profile_end()
        sustained = (sus, sus_stats)

    # ---- N > 1: the exchange step A/B (every algorithm x wire format the driver in use offers, ~10 steps each; `value` above came
    # from the DEFAULT alone) and the device time of each phase of the default, per rank (tfk_comm_timing)
    exchange_ab = exchange_phases = exchange_info = None
    if dp.enabled and getattr(reducer, "native", False):
        exchange_info = reducer.exchange_info()
        in_force = (exchange_info["reduce_scatter"], exchange_info["all_gather"], exchange_info["wire"])
        in_force_planes = bool(getattr(reducer, "planes", False))
        exchange_info["gather"] = "three-plane twins" if in_force_planes else ("bf16 shadow" if reducer.shadow else "fp32 parameters")
        ab_steps = max(1, int(os.environ.get("TFK_BENCH_AB_STEPS", "10")))  # (tests shorten it)
        exchange_ab = {"steps_each": ab_steps, "in_force_for_value": exchange_info, "ms_per_step": {}}
        if reducer.mode == "sharded":
            for algo in ("rccl", "direct"):
                for wire in ("fp32", "bf16"):
                    reducer.set_exchange(algo, wire)
                    timed_steps(3)
                    exchange_ab["ms_per_step"]["%s/%s" % (algo, wire)] = 1e3 * timed_steps(ab_steps) / ab_steps
            # emulated fp32: the owner-written three-plane twin rows on the wire instead of fp32 parameters + a rebuild on every rank
            # (TFK_DP_GATHER=planes; dataparallel.exchange_model prices it: `plane_gather`)
            if args.dtype == "float32":
                try:
                    reducer.set_gather(True)
                except Exception as exc:  # noqa: BLE001  (an engine without owner-written twins: the same answer on every rank)
                    exchange_ab["planes_unavailable"] = "%s: %s" % (type(exc).__name__, exc)
                else:
                    for algo in ("rccl", "direct"):
                        reducer.set_exchange(algo, "fp32")
                        timed_steps(3)
                        exchange_ab["ms_per_step"]["%s/fp32+planes" % algo] = 1e3 * timed_steps(ab_steps) / ab_steps
                    reducer.set_gather(in_force_planes)  # (switching back brings the fp32 masters home: collective)
            # how many weight matrices one collective carries (TFK_DP_BUCKET_MB; default 32 MiB from dataparallel.exchange_timeline,
            # never tuned on real links)
            reducer.set_exchange("rccl", "fp32")
            exchange_ab["ms_per_step_by_span_MiB"] = {}
            for mib in (16, 32, 64, 128):
                reducer.set_bucket_bytes(mib << 20)
                timed_steps(3)
                exchange_ab["ms_per_step_by_span_MiB"][str(mib)] = 1e3 * timed_steps(ab_steps) / ab_steps
            reducer.set_bucket_bytes(int(float(os.environ.get("TFK_DP_BUCKET_MB", str(DEFAULT_BUCKET_MB))) * (1 << 20)))
            # what TFK_DP_ALGO=auto would have chosen at attach: the library's own tuning pass (tfk_comm_tune, collective) on
            # scratch memory of the largest span's size, and the step with that choice
            reducer.set_exchange(None, "fp32")
            biggest = max(n for _, n in getattr(reducer, "last_launched", []) or eng.buckets())
            exchange_ab["auto"] = reducer.tune(biggest, 5)
            timed_steps(3)
            exchange_ab["ms_per_step"]["auto/fp32"] = 1e3 * timed_steps(ab_steps) / ab_steps
            # back to what `value` ran with
            if in_force[0] == in_force[1]:
                reducer.set_exchange(in_force[0], in_force[2])
            else:
                reducer.set_exchange(None, in_force[2])  # (a per-operation choice can only come from a tuning pass: keep it)
            timed_steps(3)
        reducer.timing_begin()
        timed_steps(10)
        phases, n_timed = reducer.timing_read()
        v = torch.tensor([phases[k] for k in reducer.PHASES], dtype=torch.float64, device="cuda")
        allv = [torch.zeros_like(v) for _ in range(world)]
        dist.all_gather(allv, v)
        exchange_phases = {"steps": n_timed, "unit": "device ms per step, one list entry per rank",
                           "exchange": reducer.exchange_info(),
                           "per_rank": {k: [float(a[i].item()) for a in allv] for i, k in enumerate(reducer.PHASES)},
                           "note": "timing events on the stream each phase runs on (tfk_comm_timing; every record costs its stream a "
                                   "few microseconds, so these steps are slower than `value`'s): reduce_scatter / all_reduce / "
                                   "all_gather / twin_rebuild = device time of the operation itself (mostly hidden under backward / "
                                   "the next forward pass); tail_exposed = engine-stream time between the last backward kernel and "
                                   "the first Adam kernel; gather_exposed = engine-stream time the forward pass waited for gathers"}
    elif dp.enabled:
        exchange_ab = {"unavailable": "the exchange runs through torch.distributed (dataparallel.BucketReducer): one algorithm (the "
                                      "backend's own collectives), fp32 wire; the A/B needs the in-library exchange over RCCL",
                       "ms_per_step": {"torch.distributed/fp32": 1e3 * timed_steps(min(10, args.steps)) / min(10, args.steps)}}

    if rank == 0:
        gemms = [s for s in stats if s["name"].startswith("gemm_")]
        dom = max(gemms, key=lambda s: s["total_ms"])
        achieved = dom["flops"] / dom["total_ms"] / 1e9  # TFLOP/s
        all_gemm_tf = sum(s["flops"] for s in gemms) / sum(s["total_ms"] for s in gemms) / 1e9
        value = world * T * args.steps / elapsed
        peak = w.peak
        traffic, traffic_meta, traffic_src = measured_traffic(w, kernel_label(w, dom["name"]))
        num_params = eng.buckets()[-1][0]
        # SURVEY 8d "secondary HBM bound": parameter-side bytes per step = 40 P (fp32: W read fwd + bwd, dW written, Adam 16 read +
        # 12 written) + the operand twins of the weights the optimiser writes (2 P_w bf16 shadow, 6 P_w three planes)
        p_w = eng.buckets()[-2][0]  # (the weight matrices come first in the arena)
        model_bytes = 40.0 * num_params + {"float32": 6.0, "bfloat16": 2.0, "float32_mfma": 0.0}[args.dtype] * p_w
        hbm = None
        if traffic_meta:
            b = traffic_meta["bytes_per_step"]
            hbm = {"bytes_per_step": b, "GBps_over_the_step": b / (1e-3 * my_ms) / 1e9,
                   "frac_of_8TBps": b / (1e-3 * my_ms) / 8e12,
                   "GBps_over_kernel_time_profiled": b / (traffic_meta["kernel_us_per_step_profiled"] * 1e-6) / 1e9,
                   "parameter_side_model_bytes": model_bytes, "ratio_to_model": b / model_bytes,
                   "note": "counter bytes of EVERY kernel of a step (fabric side of the L2s, Infinity-Cache hits included) over this "
                           "run's step time; model = 40 P + twin writes (SURVEY 8d); the ratio above 1 is activations, operand "
                           "re-reads across the eight L2s and the twins' reads"}
        out = {
            "metric": "acoustic frames/sec (train step)", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "clock_prewarm_ms": prewarm_ms,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_with_event_profiling": 1e3 * elapsed_profiled / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": Workload.DTYPE_TEXT[args.dtype],
            "data": "synthetic ark/scp/alignment files (SURVEY 8d) read through the feature reader + dispenser, %d "
                    "distinct micro-batches per rank cycled (a different one every step); random-init weights "
                    "N(0,1/sqrt(d_in)), zero output layer" % ring,
            "lib_build_id": eng.lib.tfk_build_id().decode(),
            "config": {"workload": "%s, %d frames/GPU/step, %s, Adam" % (w.text, T, Workload.ARITH_TEXT[args.dtype]),
                       "name": w.name, "frames_per_gpu": T, "global_frames": world * T,
                       "parallelism": "dp%d" % world, "flop_per_frame": w.flop_per_frame},
            "rccl_ranks": world if backend == "nccl" else 0, "dist_backend": backend,
            # what RAN, not what was asked for: the exchange mode the reducer settled on after probing the backend and
            # the torch.distributed calls of the last timed step, in launch order
            "exchange": reducer.mode if reducer else None,
            "exchange_driver": (("library (csrc/exchange.hip, %s)" % reducer.backend) if getattr(reducer, "native", False)
                                else "torch.distributed (dataparallel.BucketReducer)%s" % (
                                    "; in-library exchange FAILED: %s" % reducer.fallback_reason
                                    if getattr(reducer, "fallback_reason", None) else "")) if reducer else None,
            "exchange_requested": (args.exchange or os.environ.get("TFK_DP_EXCHANGE", "sharded")) if reducer else None,
            "collectives_last_step": list(reducer.last_executed) if reducer else None,
            "collective_spans_last_step": [list(x) for x in reducer.last_launched] if reducer else None,
            "dp_host_ms_per_step": ({k: 1e3 * v / max(1, reducer.host_calls["finish_and_apply"])
                                     for k, v in reducer.host_s.items()} if reducer else None),
            "per_rank_ms_per_step": per_rank_ms,
            "roofline": {"bound": "mfma", "kernel": kernel_label(w, dom["name"]), "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
                         "launches": dom["launches"], "avg_launch_us": 1e3 * dom["total_ms"] / dom["launches"],
                         "avg_launch_us_rocprofv3": rocprof_reference(w),
                         "all_gemm_tflops": all_gemm_tf,
                         "step_tflops": value / world * w.flop_per_frame / 1e12,
                         "step_frac": value / world * w.flop_per_frame / 1e12 / peak,
                         "peak_note": ("2500 / 6: six bf16 MFMAs per fp32 product.  Over operands with random significands the matrix "
                                       "pipe is POWER-bound on this chip: same cycles per launch as over zeros, clock 2.4 -> 1.87 GHz "
                                       "(profiles/r05_gemm_f32x3_clock.txt); nothing but bf16 MFMAs from registers sustains 0.72 of 2.5 PF under the "
                                       "socket's 1400 W limit (profiles/r05_mfma_bf16_energy.txt)"
                                       if args.dtype == "float32" else
                                       ("dense bf16 operands run this socket into its 1400 W limit at ~1.09 PFLOP/s (1.77 GHz), this kernel and "
                                        "the vendor library's alike; 1.5 PF at 2.39 GHz over zeros (profiles/r05_gemm_bf16_power_smi.txt)"
                                        if args.dtype == "bfloat16" else None)),
                         "hbm": hbm},
            "exchange_algorithm": exchange_info, "exchange_ab": exchange_ab, "exchange_phases": exchange_phases,
            "host_fed_value": world * T * args.steps / elapsed_host,
            "host_fed_note": "same step, micro-batch handed over as HOST numpy [%d, 440] + targets through "
                             "tfk_accumulate (PCIe inclusive; never `value`)" % T,
            "loss_first_last": [losses[0], losses[-1]],
            "loss_trace_gpu": losses[:TRACE_STEPS],
            "kernel_ms_per_step": {kernel_label(w, s["name"]): s["total_ms"] / args.steps for s in stats},
        }
        if sustained:
            sus, sus_stats = sustained
            sus["value"] = world * T / (1e-3 * sus["ms_per_step_last_window"])
            sus["unit"] = "frames/s over the last window of the leg (slowest rank)"
            sus["value_over_sustained"] = value / sus["value"]
            sg = [x for x in sus_stats if x["name"].startswith("gemm_")]
            if sg:
                sd = max(sg, key=lambda x: x["total_ms"])
                stf = sd["flops"] / sd["total_ms"] / 1e9
                sus["roofline"] = {"kernel": kernel_label(w, sd["name"]), "avg_launch_us": 1e3 * sd["total_ms"] / sd["launches"],
                                   "achieved": stf, "peak": peak, "unit": "TFLOP/s", "frac": stf / peak,
                                   "note": "20 event-bracketed steps right behind the leg (the chip still in its sustained state)"}
            sus["note"] = ("the contract's timed region is %d steps (%.0f ms) after a %.0f ms pre-warm on the same pipe; this leg runs the "
                           "same step() back to back for %.1f s -- what the rate becomes once the socket sits at its power limit"
                           % (args.steps, 1e3 * elapsed, prewarm_ms, sus["seconds"]))
            out["sustained"] = sus
        # What the exchange step should cost on 2 / 4 / 8 GPUs, predicted from THIS rank's measured step (dataparallel.
        # exchange_model): bytes on the wire per rank, link-rate time over point-to-point xGMI (direct = all peers at once, ring =
        # one link's rate), the backward time left to overlap with once the first span is ready, the step time that follows.  A
        # scaling line measured on hardware is to be held against this; at N > 1 the model uses the spans that really ran.
        from tfkaldi_amd.dataparallel import exchange_model, exchange_timeline_sweep
        per = {s["name"]: s["total_ms"] / args.steps for s in stats}
        fam = lambda *keys: sum(v for k, v in per.items() if any(q in k for q in keys))  # noqa: E731
        fwd_ms = fam("_nn(", "act_forward", "bn_stats", "softmax_xent", "loss_reduce")
        bwd_ms = fam("_nt(", "_tn(", "_dual(", "hidden_backward", "colsum")
        adam_ms = fam("adam_apply")
        step_single = my_ms
        if world > 1:  # reconstruct the single-rank step this job's ranks would run alone: full Adam, no exchange
            if (reducer.mode if reducer else "") == "sharded":
                adam_ms *= world
            step_single = fwd_ms + bwd_ms + adam_ms + fam("bn_ema_apply", "misc")
        shadow_gather = args.dtype == "bfloat16"  # (mixed precision gathers the bf16 shadow: 2 B per parameter)
        mode = (reducer.mode if reducer else (args.exchange or os.environ.get("TFK_DP_EXCHANGE", "sharded")))
        min_bytes = int(float(os.environ.get("TFK_DP_BUCKET_MB", str(DEFAULT_BUCKET_MB))) * (1 << 20))
        wire = "bf16" if os.environ.get("TFK_DP_WIRE") == "bf16" else "fp32"
        out["exchange_wire"] = getattr(reducer, "wire", "fp32") if reducer else None
        # emulated fp32 under the sharded exchange: the three-plane twins are rebuilt from the gathered fp32 parameters (4 B read,
        # 6 B written per weight; priced at 5 TB/s = 52 us at cfg2, what one RCCL rank measures: profiles/r05_dp_overhead.txt)
        n_weights = sum(n for _, n in eng.buckets()[:len(eng.buckets()) - 2])
        twin_rebuild_ms = n_weights * 10.0 / 5e12 * 1e3 if (args.dtype == "float32" and mode == "sharded") else 0.0
        out["exchange_model"] = {
            "note": "predicted from one rank's measured kernel times; weak scaling (the step of every rank is this step)",
            "per_world": {str(n): exchange_model(eng.buckets(), n, fwd_ms, bwd_ms, adam_ms, step_single, mode=mode, min_bytes=min_bytes,
                                                 gather_elem_bytes=2 if shadow_gather and mode == "sharded" else 4,
                                                 reduce_elem_bytes=2 if wire == "bf16" and mode == "sharded" else 4,
                                                 twin_rebuild_ms=twin_rebuild_ms)
                          for n in ([world] if world > 1 else [2, 4, 8])}}
        if mode == "sharded":
            # the same question as a timeline -- a fixed latency per collective, every layer of the next forward pass waiting for
            # the WHOLE gather that covers it -- over span sizes x wire rates x latencies: what the default span size was chosen from
            # (dataparallel.exchange_timeline; the wire rate RCCL reaches on the mesh is the unknown, so it is a range)
            out["exchange_model"]["timeline"] = {
                str(n): exchange_timeline_sweep(eng.buckets(), n, fwd_ms, bwd_ms, adam_ms, step_single,
                                                gather_elem_bytes=2 if shadow_gather else 4,
                                                reduce_elem_bytes=2 if wire == "bf16" else 4)
                for n in ([world] if world > 1 else [2, 4, 8])}
        if world > 1:
            m = out["exchange_model"]["per_world"][str(world)]
            out["exchange_model"]["measured_ms_per_step"] = 1e3 * elapsed / args.steps
            out["exchange_model"]["measured_over_predicted_direct"] = (1e3 * elapsed / args.steps) / m["predicted_ms_per_step_direct"]
            # the rate the wire really ran at, to hold against the model's link rates: bytes a rank sends (= receives) per step in a
            # phase (wire_bytes_per_rank_per_step) over the DEVICE time of that phase's collectives (exchange_phases: the
            # operations themselves, whatever of them was hidden), per rank; the slowest rank is the one the step waits for
            try:
                if exchange_phases and exchange_phases.get("per_rank"):
                    wb = m["wire_bytes_per_rank_per_step"]
                    rates = {}
                    for phase, key in (("reduce_scatter", "reduce_scatter_in_out"), ("all_gather", "all_gather_in_out"),
                                       ("all_reduce", "all_reduce_in_out")):
                        ms = exchange_phases["per_rank"].get(phase)
                        if key in wb and wb[key] > 0 and ms and min(ms) > 0:
                            rates[phase] = {"GB_per_s_per_rank": [wb[key] / (t * 1e-3) / 1e9 for t in ms],
                                            "slowest_rank_GB_per_s": wb[key] / (max(ms) * 1e-3) / 1e9, "bytes_per_rank": wb[key]}
                    exchange_phases["measured_wire_rate"] = dict(
                        rates, note="bytes one rank sends in the phase (fp32 gradients / gathered parameters of the spans in force for "
                                    "`value`) / device time of the phase's collectives on that rank; exchange_model's direct figure "
                                    "moves one sub-span per link to all world - 1 peers at once at link_GBps_per_direction each, its ring figure runs "
                                    "at one link's rate")
            except Exception as exc:  # noqa: BLE001  (a diagnostic must never cost the line)
                exchange_phases["measured_wire_rate"] = {"unavailable": "%s: %s" % (type(exc).__name__, exc)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], trace_cpu = cpu_baseline(w, batches, hidden)
            out["loss_trace_cpu"] = trace_cpu
            n = min(len(trace_cpu), len(out["loss_trace_gpu"]))
            if n:
                out["loss_trace_max_rel_diff"] = max(abs(a - b) / max(abs(b), 1e-30)
                                                     for a, b in zip(out["loss_trace_gpu"][:n], trace_cpu[:n]))
            if not args.no_f64_trace and args.dtype.startswith("float32") and w.name == "cfg2":
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
            out["posterior_max_err"] = posterior_error(w, eng, batches[0][0][:w.utt_len])
        if world == 1 and not args.no_decode:
            out["decode"] = decode_leg(w, eng, batches)
        if world == 1 and not args.no_eval:
            out["eval"] = eval_leg(w, eng, batches)
        if world == 1 and args.dtype.startswith("float32") and not args.no_other:
            other = "float32_mfma" if args.dtype == "float32" else "float32"
            out["exact_fp32" if other == "float32_mfma" else "emulated_fp32"] = other_arithmetic_leg(
                w, other, batches, hidden, args.steps, args.warmup, local_rank, out.get("loss_trace_f64"))
    if rank == 0:
        final["line"] = out
    eng.close()
    if not args.no_api_fed and os.environ.get("TFK_BENCH_API_FED", "1") != "0":
        try:
            fed = api_fed_leg(w, world, min(max(args.steps, 20), 60))
        except Exception as exc:  # noqa: BLE001  (the contract's line must still be printed)
            fed = {"api_fed_value": None, "api_fed_error": "%s: %s" % (type(exc).__name__, exc)}
        if rank == 0:
            out.update(fed)
    if dp.enabled:
        dist.destroy_process_group()
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)  # C stdio buffers still point at the diverted fd
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
