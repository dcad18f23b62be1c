"""The widened boundary end to end, without leaving HBM: wav samples -> fbank features (csrc/features.hip) -> CMVN + splice ->
DNN posteriors (tfk_posteriors_raw with TFK_RAW_DEVICE).  The device-resident route must give bit for bit what the route
through host numpy gives, and both must agree with the float64 oracles of the two halves chained on the CPU."""
import numpy as np
import pytest

from oracle import feat_oracle as fo
from util import make_pair

pytestmark = pytest.mark.gpu

CONF = dict(winlen='0.025', winstep='0.01', nfilt='40', nfft='512', lowfreq='0', highfreq='-1', preemph='0.97',
            include_energy='False', snip_edges='True')
C = 2
KW = dict(input_dim=40 * (2 * C + 1), num_layers=2, num_units=64, output_dim=23, nonlin="relu", batch_norm=True,
          init_learning_rate=1e-3, num_steps=10)


def _signals(rng):
    rate = 16000
    out = []
    for n in (5200, 16000, 2400, 8123):
        t = np.arange(n) / rate
        out.append(np.round(2500 * np.sin(2 * np.pi * rng.uniform(100, 2000) * t) + 500 * rng.standard_normal(n)).astype(np.int16))
    return out, rate


def test_wav_to_posteriors_on_device_equals_host_route(gpu):
    import torch
    from tfkaldi_amd import features
    from tfkaldi_amd.processing import feat
    from tfkaldi_amd.processing.feature_reader import apply_cmvn, cmvn_params, splice
    rng = np.random.default_rng(3)
    sigs, rate = _signals(rng)
    comp = feat.FeatureComputer("fbank", "nodelta", CONF)
    plan = comp.plan(rate)
    prepared = [comp._prepared(s, rate) for s in sigs]
    packed = plan.pack(prepared)
    dev_feats = plan.compute_device(packed, np.float32)               # [frames, 40] float32, stays in HBM
    host_feats = comp.compute_batch(sigs, rate, dtype=np.float32)     # the same through host numpy
    lens = [m.shape[0] for m in host_feats]
    assert dev_feats.is_cuda and dev_feats.shape == (sum(lens), 40)
    assert np.array_equal(dev_feats.cpu().numpy(), np.concatenate(host_feats))
    # one "speaker" for all four utterances: statistics from the device kernel, table as the feature reader builds it
    stats = features.cmvn_stats([host_feats])[0].astype(np.float32)
    mean, std = cmvn_params(stats)
    table = np.stack([np.stack([mean, std])] * len(lens)).astype(np.float32)

    eng, oracle = make_pair(np.random.default_rng(1), **KW)
    post_dev = eng.posteriors_raw(dev_feats, lens, C, cmvn=table)
    post_host = eng.posteriors_raw(np.concatenate(host_feats), lens, C, cmvn=table)
    assert np.array_equal(post_dev, post_host)                        # bit for bit
    # a strided view (features inside a wider matrix) works as well
    wide = torch.zeros((sum(lens), 64), dtype=torch.float32, device=dev_feats.device)
    wide[:, 8:48] = dev_feats
    assert np.array_equal(eng.posteriors_raw(wide[:, 8:48], lens, C, cmvn=table), post_host)
    # the CPU chain: reference-semantics features (float64 -> the ark's float32), host CMVN + splice, float64 DNN oracle
    ref_feats = [fo.compute_features(s, rate, "fbank", "nodelta", CONF).astype(np.float32) for s in sigs]
    X = np.concatenate([splice(apply_cmvn(m, stats), C) for m in ref_feats])
    want = oracle.posteriors(X)
    assert np.allclose(post_dev, want, rtol=2e-3, atol=1e-6), np.abs(post_dev - want).max()

    # training from device-resident features: the same loss as from the host copy, bit for bit
    y = rng.integers(0, KW["output_dim"], size=sum(lens)).astype(np.int32)
    twin, _ = make_pair(np.random.default_rng(1), **KW)
    for _ in range(2):
        eng.accumulate_raw(dev_feats, y, lens, C, cmvn=table)
        twin.accumulate_raw(np.concatenate(host_feats), y, lens, C, cmvn=table)
        assert eng.apply() == twin.apply()
    with pytest.raises(ValueError, match="float32"):
        eng.posteriors_raw(dev_feats.double(), lens, C)
    eng.close(); twin.close()
