"""GPU tool: throughput of the feature computation (csrc/features.hip) on an AURORA4-shaped batch -- 16 kHz int16
utterances of ~7 s, 40-dim log-mel filterbank features (config_AURORA4.cfg:56-76), float32 out -- in acoustic frames/s:
  * kernels only, signals resident in HBM (HIP events on the launch stream),
  * host to host (pinned-free numpy in, numpy out: PCIe both ways included),
  * the float64 numpy oracle (oracle/feat_oracle.py) on a bounded sample, as the CPU figure beside it,
and the same for the 13-MFCC + delta-delta GMM features.  One JSON line per configuration.
usage: python tools/feature_bench.py [n_utts] [out.jsonl]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import feat_oracle as fo  # noqa: E402  (the CPU figure: tools may time the oracle, the product never imports it)
from tfkaldi_amd.processing import feat  # noqa: E402

DNN = dict(winlen='0.025', winstep='0.01', nfilt='40', nfft='512', lowfreq='0', highfreq='-1', preemph='0.97',
           include_energy='False', snip_edges='True')
GMM = dict(DNN, nfilt='23', numcep='13', ceplifter='22')
CASES = [("fbank40", "fbank", "nodelta", DNN), ("mfcc13+dd", "mfcc", "ddelta", GMM),
         ("ssc40+d", "ssc", "delta", DNN)]


def flops_per_frame(nfft, nfilt, ftype, numcep):
    """algorithmic float64 operations of one frame: half-length complex transform 5 (N/2) log2(N/2), untangling the real
    spectrum ~16 per bin, power 4 per bin, the filterbank's non-zeros (each bin feeds two triangles) 2 * 2 per bin,
    (+ the DCT 2 nfilt numcep); logs / the deltas are not counted"""
    n2, bins = nfft // 2, nfft // 2 + 1
    f = 5.0 * n2 * np.log2(n2) + 16.0 * bins + 4.0 * bins + 4.0 * bins
    if ftype == "ssc":
        f += 6.0 * bins
    if ftype == "mfcc":
        f += 2.0 * nfilt * numcep
    return f


def lds_bytes_per_frame(nfft):
    """LDS traffic of the transform as written (Stockham radix 4, first pass fed from global memory, pass twiddles through
    the L1): every pass writes the N/2 complex points and every pass but the first reads them; the untangling pass reads
    the points and N/4 twiddles and writes the half spectrum; the filterbank reads every bin for two triangles (value +
    weight)"""
    n2 = nfft // 2
    passes = (int(np.log2(n2)) + 1) // 2
    data = 16.0 * n2 * (2 * passes - 1)
    untangle = 16.0 * n2 + 8.0 * n2 + 8.0 * (n2 + 1)
    return data + untangle + 32.0 * (n2 + 1)


LDS_PEAK = 256.0 * 256 * 2.4e9  # ds_read_b64 / b128: 256 B/clk/CU (MI355X_MICROARCH.md, LDS); stores reach ~80 B/clk/CU


def prepare_data_leg(n_utts, rate=16000):
    """wav.scp -> feats.ark + cmvn.ark through processing/prepare_data.py on files under /tmp: the whole host path
    (wav parsing, packing, PCIe both ways, ark writing) around the kernels"""
    import shutil
    import tempfile
    import scipy.io.wavfile as wav
    from tfkaldi_amd.processing import prepare_data
    d = tempfile.mkdtemp(prefix="tfk_featbench_", dir="/tmp")
    rng = np.random.default_rng(4)
    utts, samples = [], 0
    for i in range(n_utts):
        n = int(rng.integers(4 * rate, 10 * rate))
        uid = "spk%02d_utt%04d" % (i % 16, i)
        wav.write(os.path.join(d, uid + ".wav"), rate, np.round(2500 * rng.standard_normal(n)).astype(np.int16))
        utts.append(uid)
        samples += n
    utts.sort()
    open(os.path.join(d, "wav.scp"), "w").write("".join("%s %s\n" % (u, os.path.join(d, u + ".wav")) for u in utts))
    open(os.path.join(d, "utt2spk"), "w").write("".join("%s %s\n" % (u, u[:5]) for u in utts))
    spk = {}
    for u in utts:
        spk.setdefault(u[:5], []).append(u)
    open(os.path.join(d, "spk2utt"), "w").write("".join("%s %s\n" % (s, " ".join(us)) for s, us in sorted(spk.items())))
    open(os.path.join(d, "text"), "w").write("".join("%s A\n" % u for u in utts))
    f = os.path.join(d, "feats")
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        prepare_data.prepare_data(d, f, DNN, "fbank", "nodelta")
        t1 = time.perf_counter()
        prepare_data.compute_cmvn(f)
        t2 = time.perf_counter()
    frames = sum(1 for _ in open(os.path.join(f, "feats.scp")))
    size = os.path.getsize(os.path.join(f, "feats.ark"))
    shutil.rmtree(d)
    return {"workload": "prepare_data + compute_cmvn, fbank40, %d wav files" % n_utts, "audio_hours": samples / rate / 3600.0,
            "prepare_data_s": t1 - t0, "compute_cmvn_s": t2 - t1, "utterances": frames, "feats_ark_bytes": size,
            "frames_per_s": size / 160.0 / (t1 - t0), "times_real_time": samples / rate / (t1 - t0)}


def main():
    n_utts = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1024
    outfile = [a for a in sys.argv[1:] if not a.isdigit()]
    rng = np.random.default_rng(3)
    rate = 16000
    lens = rng.integers(4 * rate, 10 * rate, size=n_utts)
    sigs = [np.round(2500 * rng.standard_normal(n)).astype(np.int16) for n in lens]
    audio_s = float(lens.sum()) / rate
    lines = []
    for name, ftype, dyn, conf in CASES:
        comp = feat.FeatureComputer(ftype, dyn, conf)
        plan = comp.plan(rate)
        prepared = [comp._prepared(s, rate) for s in sigs]
        packed = plan.pack(prepared)
        frames = packed.n_frames
        for _ in range(3):
            out = plan.compute_device(packed, np.float32)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            out = plan.compute_device(packed, np.float32)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        t0 = time.perf_counter()
        host = comp.compute_batch(sigs, rate)
        t_host = time.perf_counter() - t0
        assert sum(h.shape[0] for h in host) == frames
        # the CPU figure: the float64 numpy restatement on a bounded sample (about 10 s of work)
        sample, t_cpu, cpu_frames = 0, 0.0, 0
        t0 = time.perf_counter()
        while sample < n_utts and time.perf_counter() - t0 < 10.0:
            ref = fo.compute_features(sigs[sample], rate, ftype, dyn, conf)
            cpu_frames += ref.shape[0]
            sample += 1
        t_cpu = time.perf_counter() - t0
        err = float(np.abs(host[0].astype(np.float64) - fo.compute_features(sigs[0], rate, ftype, dyn, conf)).max())
        fl = flops_per_frame(int(conf['nfft']), int(conf['nfilt']), ftype, int(conf.get('numcep', 0)))
        lds = lds_bytes_per_frame(int(conf['nfft']))
        fps = frames / (ms * 1e-3)
        line = {
            "metric": "acoustic frames/sec (feature computation)", "config": {"workload": name, "utterances": n_utts,
                                                                              "audio_hours": audio_s / 3600, "frames": frames, "dim": plan.dim},
            "value": fps, "unit": "frames/s", "dtype": "f64", "ms_per_batch": ms, "times_real_time": fps / 100.0,
            "host_to_host_value": frames / t_host, "host_to_host_s": t_host,
            "roofline": {"bound": "lds", "achieved": fps * lds / 1e12, "peak": LDS_PEAK / 1e12, "unit": "TB/s",
                         "frac": fps * lds / LDS_PEAK, "lds_bytes_per_frame": lds,
                         "fp64_tflops": fps * fl / 1e12, "fp64_peak_tflops": 78.6, "flop_per_frame": fl,
                         "hbm_bytes_per_frame": 2.0 * int(plan.cfg.frame_step) + 4.0 * plan.dim},
            "cpu_baseline": {"value": cpu_frames / t_cpu, "unit": "frames/s", "cores": 1, "kind": "port",
                             "sample": "%d utterances (%d frames) through oracle/feat_oracle.py (numpy float64)" % (sample, cpu_frames)},
            "max_abs_err_vs_oracle_first_utt": err,
        }
        print(json.dumps(line), flush=True)
        lines.append(line)
    line = prepare_data_leg(min(n_utts, 256))
    print(json.dumps(line), flush=True)
    lines.append(line)
    if outfile:
        with open(outfile[0], "w") as f:
            for line in lines:
                f.write(json.dumps(line) + "\n")


if __name__ == "__main__":
    main()
