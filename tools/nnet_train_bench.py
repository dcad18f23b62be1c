"""GPU tool: frames/s of `Nnet.train` ITSELF -- the reference's entry point (main.py -> nnet.train(dispenser),
reference neuralNetworks/nnet.py:80-244) -- on a synthetic corpus read through the product's ark reader, feature
reader and batch dispenser.  Everything between two optimiser steps is inside the clock: the dispenser, the host
micro-batch construction, PCIe, the engine, the printed loss line.

    python tools/nnet_train_bench.py [cfg2|cfg3|cfg4] [--batch-utts 128] [--steps 40] [--feed packed|lists|both]
    python -m torch.distributed.run --nproc-per-node N ... tools/nnet_train_bench.py   (one rank per GPU)

`measure()` is what bench.py's `api_fed_value` calls.  The clock: a trainer subclass stamps the entry of every update
call and its return; the rate runs from the entry of step `warmup` to the return of the last step (max over ranks when
data parallel).
"""
import argparse
import configparser
import contextlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# name -> (hidden layers, units, pdfs, dropout keep, compute dtype): BASELINE.json configs[1..3]
MODELS = {"cfg2": (6, 2048, 2000, 1.0, "float32"), "cfg3": (6, 2048, 4000, 1.0, "bfloat16"),
          "cfg4": (8, 4096, 8000, 0.5, "bfloat16")}
F_RAW, CONTEXT, UTT_LEN = 40, 5, 64


def _conf(expdir, model, batch_utts, per_minibatch, epochs, packed, compute_dtype=None, valid_batches=0, valid_frequency=1000000):
    layers, units, _, keep, dtype = MODELS[model]
    dtype = compute_dtype or dtype
    c = configparser.ConfigParser()
    c.add_section("directories")
    c.set("directories", "expdir", expdir)
    c.add_section("nnet")
    for k, v in dict(name="net", context_width=str(CONTEXT), num_hidden_units=str(units), num_hidden_layers=str(layers),
                     add_layer_period="0", starting_step="0", nonlin="relu", l2_norm="False", dropout=str(keep),
                     batch_norm="True", num_epochs=str(epochs), initial_learning_rate="0.001", learning_rate_decay="1",
                     batch_size=str(batch_utts), numutterances_per_minibatch=str(per_minibatch),
                     valid_batches=str(valid_batches), valid_frequency=str(valid_frequency), valid_adapt="False",
                     valid_retries="1", check_freq="1000000",
                     visualise="False", compute_dtype=dtype, packed_feed=str(bool(packed))).items():
        c.set("nnet", k, v)
    return c


def measure(model="cfg2", batch_utts=128, per_minibatch=16, steps=40, warmup=8, packed=True, workdir=None,
            utt_len=UTT_LEN, compute_dtype=None, valid_batches=0, valid_frequency=1000000):
    """Run Nnet.train for `steps` optimiser steps of `batch_utts` utterances x `utt_len` frames; returns a dict with
    frames/s over the steps after `warmup` (whole job: all ranks' frames / slowest rank's time).  valid_batches > 0: a
    held-out set of that many batches is evaluated every `valid_frequency` steps (nnet.py:168-207) INSIDE the clock; the rate
    still counts training frames only."""
    from tfkaldi_amd import synthetic
    from tfkaldi_amd.dataparallel import init_from_env
    from tfkaldi_amd.neuralNetworks import nnet as nnet_mod
    from tfkaldi_amd.processing import batchdispenser, feature_reader, target_coder
    rank, world, _ = init_from_env()
    pdfs = MODELS[model][2]
    stamps, ends = [], []

    class Timed(nnet_mod.CrossEnthropyTrainer):
        # stamps[i] = entry of the i-th update call; ends[i] = its return (the loss is on the host: the step is over).
        # The rate runs from the entry of step `warmup` to the return of the last step: every dispenser call, print and
        # schedule decision between two updates is inside, the final model save is not.
        def update(self, *a, **k):
            stamps.append(time.perf_counter())
            loss = super().update(*a, **k)
            ends.append(time.perf_counter())
            return loss

        def update_packed(self, *a, **k):
            stamps.append(time.perf_counter())
            loss = super().update_packed(*a, **k)
            ends.append(time.perf_counter())
            return loss

    with contextlib.ExitStack() as stack:
        if workdir is None:
            workdir = stack.enter_context(tempfile.TemporaryDirectory(prefix="tfkaldi_nnet_bench_"))
        corpus = os.path.join(workdir, "corpus_rank%d" % rank)  # (same seed on every rank: identical files)
        # ArkReader.split drops the LAST scp entry as the reference does (ark.py:161-165): one spare utterance
        paths = synthetic.write_corpus(corpus, batch_utts * (steps + valid_batches) + 1, pdfs, feat_dim=F_RAW, utt_len=utt_len)
        reader = feature_reader.FeatureReader(paths["feats_scp"], paths["cmvn_scp"], paths["utt2spk"], CONTEXT, utt_len)
        coder = target_coder.AlignmentCoder(lambda x, y: x, pdfs)
        disp = batchdispenser.AlignmentBatchDispenser(reader, coder, batch_utts, paths["alignments"])
        net = nnet_mod.Nnet(_conf(os.path.join(workdir, "exp_rank%d" % rank), model, batch_utts, per_minibatch, 1, packed,
                                  compute_dtype, valid_batches, valid_frequency), F_RAW, pdfs)
        original = nnet_mod.CrossEnthropyTrainer
        nnet_mod.CrossEnthropyTrainer = Timed
        try:
            with open(os.devnull, "w") as sink, contextlib.redirect_stdout(sink):
                net.train(disp)
        finally:
            nnet_mod.CrossEnthropyTrainer = original
    done = len(ends)
    if done <= warmup:
        raise RuntimeError("Nnet.train ran %d steps, not more than the %d of warm-up" % (done, warmup))
    elapsed = ends[-1] - stamps[warmup]
    if world > 1:
        import torch
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    frames = (done - warmup) * batch_utts * utt_len
    return {"value": frames / elapsed, "unit": "frames/s", "ms_per_step": 1e3 * elapsed / (done - warmup),
            "steps": done - warmup, "warmup": warmup, "model": model, "feed": "packed" if packed else "lists",
            "frames_per_step": batch_utts * utt_len, "microbatches_per_step": batch_utts // per_minibatch,
            "n_gpus": world}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model", nargs="?", default="cfg2", choices=sorted(MODELS))
    ap.add_argument("--batch-utts", type=int, default=128)
    ap.add_argument("--per-minibatch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--feed", choices=["packed", "lists", "both"], default="both")
    args = ap.parse_args()
    for feed in (["packed", "lists"] if args.feed == "both" else [args.feed]):
        out = measure(args.model, args.batch_utts, args.per_minibatch, args.steps, args.warmup, packed=feed == "packed")
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps(out))


if __name__ == "__main__":
    main()
