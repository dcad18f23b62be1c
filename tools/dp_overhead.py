"""GPU tool (one GPU): what the exchange MACHINERY costs the host and the step, measured with a single-rank RCCL group
(TFK_FORCE_DP=1: every collective runs through RCCL on one rank, so the wire time is ~0 and what remains is the host
path -- ctypes bucket / layer callbacks, torch.distributed launches, per-span tfk_apply_span calls, stream waits).

For cfg2 (fp32) and cfg3 per GPU (bf16): ms/step of the plain step, of `allreduce`, and of `sharded`, with the host time
spent inside on_bucket / on_layer / finish_and_apply per step (perf_counter inside the reducer) and the collectives issued.
The reference seam is neuralNetworks/trainer.py:165-169 (G += g) and :174-184 (apply).

    python tools/dp_overhead.py > profiles/rNN_dp_overhead.txt
Each configuration runs in its own process (one process group per process).
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {  # T per GPU, F, L, H, O, keep_prob, dtype
    "cfg2": (1024, 440, 6, 2048, 2000, 1.0, "float32"),
    "cfg3": (1024, 440, 6, 2048, 4000, 1.0, "bfloat16"),
    "cfg4": (2048, 440, 8, 4096, 8000, 0.5, "bfloat16"),
}


def child(name, mode):
    import numpy as np
    import torch
    from tfkaldi_amd import _lib
    from tfkaldi_amd.dataparallel import DataParallel, init_from_env
    from tfkaldi_amd.engine import Engine
    T, F, L, H, O, keep, dtype = CONFIGS[name]
    if mode != "plain":
        os.environ["TFK_FORCE_DP"] = "1"
        os.environ.setdefault("MASTER_PORT", "29533")
        # "allreduce" / "sharded": the exchange launched by the library (csrc/exchange.hip, the default on RCCL);
        # "torch-allreduce" / "torch-sharded": the same protocol driven from Python through torch.distributed (round 3)
        os.environ["TFK_DP_COMM"] = "torch" if mode.startswith("torch-") else "native"
        # "sharded+planes" / "sharded+direct" / "sharded+direct+planes" (round 6): the owner-written three-plane twin rows gathered
        # in place of fp32 parameters + a rebuild; the direct algorithm (with one rank: empty send / recv groups + the owner's sum)
        if "+planes" in mode:
            os.environ["TFK_DP_GATHER"] = "planes"
        if "+direct" in mode:
            os.environ["TFK_DP_ALGO"] = "direct"
    init_from_env()
    dp = DataParallel(mode=None if mode == "plain" else mode.replace("torch-", "").split("+")[0])
    eng = Engine(_lib.make_config(F, L, H, O, nonlin="relu", batch_norm=True, keep_prob=keep, max_frames=T,
                                  num_steps=1000, compute_dtype=dtype),
                 torch_state=dp.enabled or bool(os.environ.get("TFK_PLAIN_TORCH_STATE")))  # (experiment: plain step on a torch stream)
    eng.init_hidden_weights(np.random.default_rng(7))
    X = torch.randn(T, F, device="cuda")
    y = torch.randint(0, O, (T,), device="cuda", dtype=torch.int32)
    torch.cuda.synchronize()
    red = dp.reducer(eng) if dp.enabled else None
    if red:
        red.begin_step(eng)

    def step():
        eng.accumulate_device(X.data_ptr(), F, y.data_ptr(), T, last=True)
        return red.finish_and_apply(eng) if red else eng.apply()

    for _ in range(8):
        step()
    eng.synchronize()
    torch.cuda.synchronize()
    if red:
        for k in red.host_s:
            red.host_s[k] = 0.0
            red.host_calls[k] = 0
    K = 60
    t_host = 0.0
    t0 = time.perf_counter()
    for _ in range(K):
        th = time.perf_counter()
        step()
        t_host += time.perf_counter() - th
    eng.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    out = {"config": name, "mode": mode, "ms_per_step": 1e3 * dt, "host_ms_per_step_in_step_call": 1e3 * t_host / K}
    if red:
        out["exchange_ran"] = red.mode
        out["driver"] = "library" if getattr(red, "native", False) else "torch.distributed"
        out["collectives_per_step"] = list(red.last_executed)
        out["host_ms_per_step"] = {k: 1e3 * v / K for k, v in red.host_s.items()}
        out["callbacks_per_step"] = {k: v / K for k, v in red.host_calls.items()}
    print("DPOVERHEAD " + json.dumps(out))
    eng.close()
    if dp.enabled:
        import torch.distributed as dist
        dist.destroy_process_group()


def main():
    if len(sys.argv) == 3:
        return child(sys.argv[1], sys.argv[2])
    rows = []
    for name in (sys.argv[1:] or ["cfg2", "cfg3", "cfg4"]):
        x3 = CONFIGS[name][6] == "float32"
        for mode in ("plain", "allreduce", "sharded") + (("sharded+planes", "sharded+direct", "sharded+direct+planes") if x3 else
                                                          ("sharded+direct",)) + ("torch-allreduce", "torch-sharded"):
            env = dict(os.environ, MASTER_ADDR="127.0.0.1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                       HSA_ENABLE_IPC_MODE_LEGACY="0")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), name, mode], env=env, capture_output=True,
                               text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("DPOVERHEAD ")]
            if not line:
                print("FAILED %s %s\n%s" % (name, mode, r.stderr[-1500:]))
                continue
            rows.append(json.loads(line[0][len("DPOVERHEAD "):]))
    print("exchange machinery on ONE rank (RCCL group of size 1: wire time ~ 0; what is left is the host path + the extra launches)")
    print("%-5s %-15s %9s %12s | host ms/step inside: %9s %9s %16s | collectives per step" % (
        "cfg", "mode", "ms/step", "host ms/step", "on_bucket", "on_layer", "finish_and_apply"))
    base = {}
    for r in rows:
        h = r.get("host_ms_per_step", {})
        if r["mode"] == "plain":
            base[r["config"]] = r["ms_per_step"]
        print("%-5s %-15s %9.3f %12.3f | %30.3f %9.3f %16.3f | %s" % (
            r["config"], r["mode"], r["ms_per_step"], r["host_ms_per_step_in_step_call"], h.get("on_bucket", 0.0),
            h.get("on_layer", 0.0), h.get("finish_and_apply", 0.0), ", ".join(
                "%dx %s" % (r.get("collectives_per_step", []).count(c), c)
                for c in sorted(set(r.get("collectives_per_step", []))))))
    for r in rows:
        if r["mode"] != "plain" and r["config"] in base:
            print("%s %s: +%.3f ms/step over the plain step (%.1f %%)" % (
                r["config"], r["mode"], r["ms_per_step"] - base[r["config"]],
                100.0 * (r["ms_per_step"] / base[r["config"]] - 1.0)))
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
