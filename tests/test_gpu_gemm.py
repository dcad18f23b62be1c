"""fp32 MFMA GEMM (tfkaldi_amd/csrc/gemm_f32.hip) against float64 numpy, through the C ABI.

Tolerance: v_mfma_f32_32x32x2_f32 is bitwise an fp32 fmaf chain, so the error vs float64 is plain fp32
round-off: |err| <= 4e-7 * sum_k |a_k b_k| + 1e-6 (cdna_hip_programming.md section 3 quotes
0.75-1.5e-7 * sum|ab| up to K = 1024).
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pad4(n):
    return (n + 3) & ~3


def _dev(torch, a, rows, cols):
    """host [rows, cols] -> device buffer with leading dimension pad4(cols), zero padded."""
    ld = _pad4(cols)
    buf = np.zeros((rows, ld), dtype=np.float32)
    buf[:, :cols] = a
    return torch.from_numpy(buf).cuda(), ld


def _run(lib, torch, layout, M, N, K, cfg, epi=0, seed=0):
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(seed)
    if layout == _lib.GEMM_NN:
        A = rng.standard_normal((M, K)); B = rng.standard_normal((K, N))
        ref, absref = A @ B, np.abs(A) @ np.abs(B)
        dA, lda = _dev(torch, A, M, K); dB, ldb = _dev(torch, B, K, N)
    elif layout == _lib.GEMM_NT:
        A = rng.standard_normal((M, K)); B = rng.standard_normal((N, K))
        ref, absref = A @ B.T, np.abs(A) @ np.abs(B).T
        dA, lda = _dev(torch, A, M, K); dB, ldb = _dev(torch, B, N, K)
    else:
        A = rng.standard_normal((K, M)); B = rng.standard_normal((K, N))
        ref, absref = A.T @ B, np.abs(A).T @ np.abs(B)
        dA, lda = _dev(torch, A, K, M); dB, ldb = _dev(torch, B, K, N)
    A32 = None
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    dC, ldc = _dev(torch, C0, M, N)
    bias = rng.standard_normal(N).astype(np.float32)
    dbias = torch.from_numpy(bias).cuda()
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.tfk_gemm_f32(ctypes.c_void_p(st), layout, ctypes.c_void_p(dA.data_ptr()), lda,
                          ctypes.c_void_p(dB.data_ptr()), ldb, ctypes.c_void_p(dC.data_ptr()), ldc, M, N, K,
                          ctypes.c_void_p(dbias.data_ptr()), epi, cfg)
    assert rc == 0, lib.tfk_last_error()
    torch.cuda.synchronize()
    out = dC.cpu().numpy()
    # operands were rounded to fp32 on the way in
    if layout == _lib.GEMM_NN:
        ref = A.astype(np.float32).astype(np.float64) @ B.astype(np.float32).astype(np.float64)
    elif layout == _lib.GEMM_NT:
        ref = A.astype(np.float32).astype(np.float64) @ B.astype(np.float32).astype(np.float64).T
    else:
        ref = A.astype(np.float32).astype(np.float64).T @ B.astype(np.float32).astype(np.float64)
    if epi & _lib.EPI_BIAS:
        ref = ref + bias
    if epi & _lib.EPI_ACCUM:
        ref = ref + C0
    if epi & _lib.EPI_RELU:
        ref = np.maximum(ref, 0)
    err = np.abs(out[:, :N] - ref)
    tol = 4e-7 * (absref + np.abs(ref)) + 1e-6
    assert (err <= tol).all(), "layout %d cfg %d %dx%dx%d: max err %g (tol %g)" % (
        layout, cfg, M, N, K, err.max(), tol[np.unravel_index(err.argmax(), err.shape)])
    # padding columns of C must be untouched
    assert (out[:, N:] == 0).all()


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("cfg", list(range(10)))
def test_gemm_ragged_shapes(gpu, layout, cfg):
    import torch
    # nothing a multiple of the tile: exercises every edge predicate and the zero-padded float4 tails
    _run(gpu, torch, layout, 197, 203, 75, cfg)
    _run(gpu, torch, layout, 70, 330, 33, cfg, seed=1)


@pytest.mark.parametrize("layout,epi", [(0, 0), (0, 1), (0, 5), (2, 0), (2, 2), (1, 0)])
def test_gemm_epilogues(gpu, layout, epi):
    import torch
    for cfg in (-1, 0, 3, 6, 9):
        _run(gpu, torch, layout, 130, 100, 64, cfg, epi=epi)      # ragged: predicated epilogue
        _run(gpu, torch, layout, 256, 256, 96, cfg, epi=epi)      # full tiles: straight-line epilogue


def test_gemm_rejects_unsupported_epilogue(gpu):
    import torch
    with pytest.raises(AssertionError):
        _run(gpu, torch, 1, 64, 64, 64, -1, epi=1)  # NT never carries a bias


@pytest.mark.parametrize("layout,M,N,K", [(0, 1024, 2048, 440), (0, 1024, 2000, 2048), (1, 1024, 2048, 2000),
                                          (2, 2048, 2000, 1024), (2, 440, 2048, 1024)])
def test_gemm_baseline_shapes(gpu, layout, M, N, K):
    """the contractions of BASELINE cfg2 (6x2048, 440 in, 2000 pdfs, 1024 frames) with the heuristic tile."""
    import torch
    _run(gpu, torch, layout, M, N, K, -1)


def test_gemm_transpose_detecting(gpu):
    """A = I with an asymmetric B: a swapped row/col in the C write cannot pass."""
    import torch
    from tfkaldi_amd import _lib
    n = 96
    A = np.eye(n, dtype=np.float32)
    B = (np.arange(n)[:, None] * 1000 + np.arange(n)[None, :]).astype(np.float32)
    dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    dC = torch.zeros((n, n), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for cfg in range(10):
        dC.zero_()
        rc = gpu.tfk_gemm_f32(ctypes.c_void_p(st), _lib.GEMM_NN, ctypes.c_void_p(dA.data_ptr()), n,
                              ctypes.c_void_p(dB.data_ptr()), n, ctypes.c_void_p(dC.data_ptr()), n, n, n, n, None, 0, cfg)
        assert rc == 0
        torch.cuda.synchronize()
        assert (dC.cpu().numpy() == B).all(), "cfg %d" % cfg


@pytest.mark.parametrize("layout", [0, 1, 2])
def test_gemm_random_shapes_lds_dma_tile(gpu, layout):
    """Random shapes through the LDS-DMA 64x64 tile (config 9, the hot path's tile) and the heuristic: extents of 1,
    K shorter than one 32-k tile, K not a multiple of 4, ragged last tiles in both directions, K long enough for
    every ring slot to be recycled many times."""
    import torch
    rng = np.random.default_rng(100 + layout)
    shapes = [(1, 1, 1), (1, 70, 5), (65, 1, 31), (64, 64, 32), (63, 65, 33), (2, 3, 700), (130, 190, 1027)]
    shapes += [tuple(int(x) for x in rng.integers(1, 260, size=3)) for _ in range(10)]
    for n, (M, N, K) in enumerate(shapes):
        _run(gpu, torch, layout, M, N, K, 9 if n % 3 else -1, seed=n)
    _run(gpu, torch, layout, 96, 80, 1500, 9, epi=(2 if layout == 2 else 0), seed=77)
