"""bf16 MFMA GEMM (tfkaldi_amd/csrc/gemm_bf16.hip) against float64 numpy on the SAME bf16-rounded operands,
through the C ABI.  Products of two bf16 values are exact in fp32, so the only error is the fp32 accumulation:
|err| <= 4e-7 * sum_k |a_k b_k| + 1e-6, the bound of the fp32 kernel's test.  Inputs are random and not symmetric,
so a transposed operand or a wrong k order inside ds_read_b64_tr_b16 / the MFMA operand cannot pass."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EPI_BIAS, EPI_ACCUM = 1, 2


def _pad(n, m):
    return (n + m - 1) // m * m


def _bf16_dev(torch, a):
    """host float [rows, cols] -> (device bf16 buffer with ld = pad8(cols) and zero padding, ld, rounded values)"""
    rows, cols = a.shape
    ld = _pad(cols, 8)
    buf = torch.zeros((rows, ld), dtype=torch.bfloat16)
    buf[:, :cols] = torch.from_numpy(a.astype(np.float32)).to(torch.bfloat16)
    rounded = buf[:, :cols].to(torch.float64).numpy()
    return buf.cuda(), ld, rounded


def _run(lib, torch, layout, M, N, K, epi=0, seed=0, cfg=-1):
    lib.tfk_gemm_bf16_force_config(cfg)
    rng = np.random.default_rng(seed)
    shapes = {0: ((M, K), (K, N)), 1: ((M, K), (N, K)), 2: ((K, M), (K, N))}[layout]
    A = rng.standard_normal(shapes[0]); B = rng.standard_normal(shapes[1])
    dA, lda, Ar = _bf16_dev(torch, A)
    dB, ldb, Br = _bf16_dev(torch, B)
    if layout == 0:
        ref, absref = Ar @ Br, np.abs(Ar) @ np.abs(Br)
    elif layout == 1:
        ref, absref = Ar @ Br.T, np.abs(Ar) @ np.abs(Br).T
    else:
        ref, absref = Ar.T @ Br, np.abs(Ar).T @ np.abs(Br)
    ldc = _pad(N, 4)
    C0 = np.zeros((M, ldc), dtype=np.float32)
    C0[:, :N] = rng.standard_normal((M, N))
    dC = torch.from_numpy(C0.copy()).cuda()
    bias = rng.standard_normal(N).astype(np.float32)
    dbias = torch.from_numpy(bias).cuda()
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.tfk_gemm_bf16(ctypes.c_void_p(st), layout, ctypes.c_void_p(dA.data_ptr()), lda,
                           ctypes.c_void_p(dB.data_ptr()), ldb, ctypes.c_void_p(dC.data_ptr()), ldc, M, N, K,
                           ctypes.c_void_p(dbias.data_ptr()), epi)
    lib.tfk_gemm_bf16_force_config(-1)
    assert rc == 0, lib.tfk_last_error()
    torch.cuda.synchronize()
    out = dC.cpu().numpy()
    if epi & EPI_BIAS:
        ref = ref + bias
        absref = absref + np.abs(bias)
    if epi & EPI_ACCUM:
        ref = ref + C0[:, :N]
        absref = absref + np.abs(C0[:, :N])
    err = np.abs(out[:, :N] - ref)
    bound = 4e-7 * absref + 1e-6
    assert (err <= bound).all(), "layout %d %dx%dx%d epi %d: max err/bound %.2f" % (
        layout, M, N, K, epi, float((err / bound).max()))
    assert (out[:, N:] == C0[:, N:]).all()  # padding columns untouched


SHAPES = [(64, 64, 64), (128, 192, 256), (1, 1, 1), (37, 29, 13), (65, 63, 130), (100, 250, 72), (256, 2000, 440),
          (1024, 2048, 2048)]


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("shape", SHAPES)
def test_gemm_bf16(gpu, layout, shape):
    import torch
    from tfkaldi_amd import _lib
    M, N, K = shape
    if layout == 2:
        M, K = K, M  # TN: the long dimension is the contraction (frames)
    _run(_lib.load(), torch, layout, M, N, K, seed=layout * 100 + M)


# every tile configuration (0-2 register-staged, 3-6 LDS-DMA staged: gemm_bf16.h) on ragged shapes that exercise
# edge tiles in m, n and k, a K shorter than the ring is deep, and one full-size problem per layout
CFG_SHAPES = [(37, 29, 13), (130, 70, 200), (300, 200, 136), (257, 129, 520), (512, 512, 1088)]


@pytest.mark.parametrize("cfg", range(9))
@pytest.mark.parametrize("layout", [0, 1, 2])
def test_gemm_bf16_every_config(gpu, layout, cfg):
    import torch
    from tfkaldi_amd import _lib
    for M, N, K in CFG_SHAPES:
        _run(_lib.load(), torch, layout, M, N, K, seed=cfg * 31 + layout, cfg=cfg)
    for epi in ((EPI_BIAS,) if layout == 0 else (EPI_ACCUM,) if layout == 2 else ()):
        _run(_lib.load(), torch, layout, 200, 136, 72, epi=epi, seed=5, cfg=cfg)


@pytest.mark.parametrize("layout,epi", [(0, EPI_BIAS), (2, EPI_ACCUM)])
def test_gemm_bf16_epilogues(gpu, layout, epi):
    import torch
    from tfkaldi_amd import _lib
    for M, N, K in ((70, 90, 200), (256, 128, 64)):
        _run(_lib.load(), torch, layout, M, N, K, epi=epi, seed=7)


# the backward pair of a layer in one launch (gemm_bf16_dual): dA = dZ . W^T (NT) and dW (+)= in^T . dZ (TN) share dZ
DUAL_SHAPES = [  # (T, d_in, d_out): NT is [T, d_in] over K = d_out, TN is [d_in, d_out] over K = T
    (1024, 2048, 2048), (1024, 2048, 4000), (300, 520, 264), (129, 257, 136), (2048, 1024, 1000)]


@pytest.mark.parametrize("shape", DUAL_SHAPES)
@pytest.mark.parametrize("epi_tn", [0, EPI_ACCUM])
def test_gemm_bf16_dual(gpu, shape, epi_tn):
    """every block geometry of the dual launch (the environment switch is read once per process, so the geometry is
    chosen by the heuristic here; tools/gemm_bf16_dual_bench.py sweeps the others with the same check)"""
    import torch
    from tfkaldi_amd import _lib
    lib = _lib.load()
    T, d_in, d_out = shape
    if lib.tfk_gemm_bf16_dual_config(T, d_in, d_in, d_out) == 0:
        pytest.skip("pair not eligible for the dual launch")
    rng = np.random.default_rng(T + d_in + d_out + epi_tn)
    dz, ld_dz, dzr = _bf16_dev(torch, rng.standard_normal((T, d_out)))
    W, ld_w, Wr = _bf16_dev(torch, rng.standard_normal((d_in, d_out)))
    X, ld_x, Xr = _bf16_dev(torch, rng.standard_normal((T, d_in)))
    ldc_a, ldc_w = _pad(d_in, 4), _pad(d_out, 4)
    dA = torch.zeros((T, ldc_a), dtype=torch.float32, device="cuda")
    G0 = np.zeros((d_in, ldc_w), dtype=np.float32)
    G0[:, :d_out] = rng.standard_normal((d_in, d_out))
    G = torch.from_numpy(G0.copy()).cuda()
    torch.cuda.synchronize()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = lib.tfk_gemm_bf16_dual(st, p(dz), ld_dz, p(W), ld_w, p(dA), ldc_a, T, d_in, d_out,
                                p(X), ld_x, p(dz), ld_dz, p(G), ldc_w, d_in, d_out, T, epi_tn)
    assert rc == 0, lib.tfk_last_error()
    torch.cuda.synchronize()
    ref_a, abs_a = dzr @ Wr.T, np.abs(dzr) @ np.abs(Wr).T
    ref_w, abs_w = Xr.T @ dzr, np.abs(Xr).T @ np.abs(dzr)
    if epi_tn:
        ref_w, abs_w = ref_w + G0[:, :d_out], abs_w + np.abs(G0[:, :d_out])
    for name, out, ref, absref in (("dA", dA.cpu().numpy()[:, :d_in], ref_a, abs_a),
                                   ("dW", G.cpu().numpy()[:, :d_out], ref_w, abs_w)):
        err = np.abs(out - ref)
        bound = 4e-7 * absref + 1e-6
        assert (err <= bound).all(), "%s %s: max err/bound %.2f" % (name, shape, float((err / bound).max()))
