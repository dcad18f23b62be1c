"""GPU tool: randomised sweep of the feature computation against the float64 oracle -- transform lengths 32..4096, 1..200
filters, every feature type / dynamic, 4-48 kHz, utterances from 1 sample up, int16, float64 and float32 input, batches of mixed
lengths.  usage: python tools/fuzz_features.py [n_configs] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import feat_oracle as fo  # noqa: E402
from tfkaldi_amd.processing import feat  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    worst = 0.0
    flips, total = {}, {}
    for it in range(n):
        rate = int(rng.choice([4000, 8000, 11025, 16000, 22050, 44100, 48000]))
        nfft = int(2 ** rng.integers(5, 13))
        winlen = float(rng.choice([0.005, 0.01, 0.02, 0.025, 0.032, 0.05]))
        winstep = float(rng.choice([0.0025, 0.005, 0.01, 0.0125, 0.02]))
        nfilt = int(min(nfft // 2, rng.choice([1, 3, 8, 13, 23, 26, 40, 64, 65, 80, 128, 200])))
        ftype = str(rng.choice(["fbank", "mfcc", "ssc"]))
        dyn = str(rng.choice(["nodelta", "delta", "ddelta"]))
        low = int(rng.choice([0, 0, 20, 100, 300]))
        high = int(rng.choice([-1, -1, rate // 2, int(rate * 0.45)]))
        conf = dict(winlen=repr(winlen), winstep=repr(winstep), nfilt=str(nfilt), nfft=str(nfft), lowfreq=str(low),
                    highfreq=str(high), preemph=str(rng.choice(["0.97", "0.95", "0", "1.0"])),
                    include_energy=str(bool(rng.integers(2))), snip_edges=str(bool(rng.integers(2))),
                    numcep=str(int(rng.integers(1, nfilt + 3))), ceplifter=str(rng.choice(["22", "0", "7.5"])))
        if low >= (high if high > 0 else rate // 2):
            continue
        lens = [int(x) for x in rng.choice([1, 2, 17, 100, 399, 400, 401, 1000, 5000, 20000], size=int(rng.integers(1, 7)))]
        as_float = int(rng.integers(3))  # 0: int16, 1: float64, 2: float32 (numpy keeps the pre-emphasis in float32 there)
        sigs = []
        for m in lens:
            x = 3000 * np.sin(2 * np.pi * rng.uniform(50, rate / 2.2) * np.arange(m) / rate) + 300 * rng.standard_normal(m)
            sigs.append(np.round(x).astype(np.int16) if as_float == 0 else (x * 1e-3).astype(np.float64 if as_float == 1 else np.float32))
        try:
            comp = feat.FeatureComputer(ftype, dyn, conf)
            got = comp.compute_batch(sigs, rate, dtype=np.float64)
        except Exception as exc:  # noqa: BLE001
            print("config %d: %s %s %s -> %s: %s" % (it, ftype, dyn, conf, type(exc).__name__, exc))
            raise
        for s, g in zip(sigs, got):
            with np.errstate(all="ignore"):
                ref = fo.compute_features(s, rate, ftype, dyn, conf)
            assert g.shape == ref.shape, (it, conf, g.shape, ref.shape)
            assert np.array_equal(np.isnan(g), np.isnan(ref)) and np.array_equal(np.isinf(g), np.isinf(ref)), (it, conf)
            ok = np.isfinite(ref)
            # ssc is a ratio of two band energies: where a band holds no energy at all (a pure tone elsewhere) both are
            # round-off and the ratio is noise in the reference as well -- compare where the reference is well-conditioned
            scale = np.maximum(np.abs(ref[ok]), 1.0)
            err = np.abs(g[ok] - ref[ok]) / scale
            tol = 1e-6 if ftype == "ssc" else 1e-8
            bad = err > tol
            if bad.any() and ftype != "ssc":
                raise AssertionError("config %d %s %s %s len %d: max rel err %.3e" % (it, ftype, dyn, conf, len(s), err.max()))
            if ftype != "ssc" and err.size:
                worst = max(worst, float(err.max()))
            # float32 comparison where the value is not itself round-off (a delta of a constant centroid, a cepstrum of a
            # flat spectrum: ~1e-13 in both, unrelated digits)
            sig_ok = ok & (np.abs(ref) > 1e-6)
            a, b = g[sig_ok].astype(np.float32), ref[sig_ok].astype(np.float32)
            if os.environ.get("TFK_FUZZ_VERBOSE") and (a != b).mean() > 0.01:
                print("  config %d %s %s len %d float=%s: %.1f%% float32 values differ, max rel err %.2e\n    %s"
                      % (it, ftype, dyn, len(s), as_float, 100 * (a != b).mean(), err.max() if err.size else 0, conf))
            flips[ftype] = flips.get(ftype, 0) + int((a != b).sum())
            total[ftype] = total.get(ftype, 0) + a.size
        comp.plan(rate).close()
    print("%d configurations: worst relative error (fbank / mfcc) %.3e; float32 values that differ from the oracle's: %s"
          % (n, worst, ", ".join("%s %d of %d" % (k, flips[k], total[k]) for k in sorted(total))))


if __name__ == "__main__":
    main()
