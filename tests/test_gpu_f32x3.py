"""TFK_DTYPE_F32X3 (`compute_dtype = "float32x3"`): fp32 arithmetic emulated on the bf16 matrix pipe -- every GEMM operand split,
exactly, into three bfloat16 planes, six plane products accumulated in fp32 (include/tfkaldi_hip.h, csrc/gemm_bf16.h:
gemm_bf16x3).  It claims to BE fp32 arithmetic, so it is held to the fp32 bounds: the stand-alone contraction to the bound the
exact-fp32 MFMA kernel is held to (tests/test_gpu_gemm.py: 4e-7 * sum|ab| + 1e-6 against float64), and the engine to the fp32
suites, run again with this arithmetic against the same float64 oracle and the same tolerances."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p8 = lambda n: (n + 7) & ~7  # noqa: E731
p4 = lambda n: (n + 3) & ~3  # noqa: E731


def _planes(lib, torch, x, ld=None):
    """the tiled three-plane twin of x (csrc/x3_layout.h) and its leading dimension"""
    from tfkaldi_amd import x3
    return x3.split(lib, x, ld)


def _gemm(lib, torch, layout, Ap, lda, Bp, ldb, C, ldc, M, N, K, bias=None, epi=0):
    from tfkaldi_amd import _lib
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.tfk_gemm_bf16x3(st, layout, ctypes.c_void_p(Ap.data_ptr()), lda, ctypes.c_void_p(Bp.data_ptr()), ldb,
                                   ctypes.c_void_p(C.data_ptr()), ldc, M, N, K,
                                   ctypes.c_void_p(bias.data_ptr()) if bias is not None else None, epi))


def _run(lib, layout, M, N, K, epi=0, seed=0, wide=False):
    """wide: leading dimensions one unit (32 columns) longer than the matrix needs"""
    import torch
    from tfkaldi_amd import x3
    g = torch.Generator(device="cuda").manual_seed(seed)
    shape_a = (K, M) if layout == 2 else (M, K)
    shape_b = (N, K) if layout == 1 else (K, N)
    A = torch.randn(*shape_a, device="cuda", generator=g) * 3
    B = torch.randn(*shape_b, device="cuda", generator=g)
    Ad, Bd = A.double(), B.double()
    ref = (Ad.T if layout == 2 else Ad) @ (Bd.T if layout == 1 else Bd)
    sab = (Ad.abs().T if layout == 2 else Ad.abs()) @ (Bd.abs().T if layout == 1 else Bd.abs())
    Ap, lda = _planes(lib, torch, A, x3.padded_ld(A.shape[1]) + 32 if wide else None)
    Bp, ldb = _planes(lib, torch, B, x3.padded_ld(B.shape[1]) + 32 if wide else None)
    for X, Xp, ld in ((A, Ap, lda), (B, Bp, ldb)):  # the split is exact, plane by plane a bf16
        r, c = X.shape
        pl = x3.planes(Xp, r, ld)
        assert torch.equal(sum(q[:, :c].float() for q in pl), X)
        assert all(bool((q[:, c:] == 0).all()) for q in pl)  # padding columns are zeros
    ldc = p4(N)
    C0 = torch.randn(M, ldc, device="cuda", generator=g)
    C = C0.clone()
    bias = torch.randn(N, device="cuda", generator=g)
    _gemm(lib, torch, layout, Ap, lda, Bp, ldb, C, ldc, M, N, K, bias, epi)
    torch.cuda.synchronize()
    want = ref + (bias.double() if epi & 1 else 0) + (C0[:, :N].double() if epi & 2 else 0)
    err = (C[:, :N].double() - want).abs()
    tol = 4e-7 * (sab + want.abs()) + 1e-6
    assert bool((err <= tol).all()), "layout %d %dx%dx%d epi %d: max err %g" % (layout, M, N, K, epi, err.max().item())
    assert bool((C[:, N:] == C0[:, N:]).all())  # padding columns untouched


@pytest.mark.parametrize("layout", [0, 1, 2])
def test_contraction_within_the_fp32_kernels_bound(gpu, layout):
    """ragged shapes (every edge predicate, K shorter than one slot, K not a multiple of 8), the BASELINE shapes, one large
    shape per block geometry (128x64, 128x128), the epilogues the layout carries"""
    epi = {0: 1, 1: 0, 2: 2}[layout]
    for n, (M, N, K) in enumerate([(197, 203, 75), (70, 330, 33), (1, 1, 1), (1, 70, 5), (65, 1, 31), (129, 257, 1027),
                                   (2, 3, 700), (1024, 2048, 440), (1024, 2000, 2048), (2048, 2048, 1024), (2048, 4096, 512),
                                   # two blocks per tile over half of K each (NN / NT): K halves of 17 + 16 and 31 + 31 ring tiles,
                                   # a ragged last tile row
                                   (1024, 2048, 1040), (1000, 2048, 1976),
                                   # the narrow layer's weight gradient (TN): 128x64 blocks, two per tile
                                   (440, 2048, 1024), (500, 2000, 1100)]):
        _run(gpu, layout, M, N, K, epi=epi if n % 2 == 0 else 0, seed=n)
    # odd row counts (the last row pair half empty), 2000 pdfs / 440 inputs (the last unit of a row part padding), wide twins
    for n, (M, N, K) in enumerate([(1023, 2000, 2047), (1024, 2048, 2000), (441, 2000, 1023), (129, 71, 199), (71, 331, 33)]):
        _run(gpu, layout, M, N, K, epi=0, seed=100 + n, wide=n % 2 == 0)


@pytest.mark.parametrize("layout", [0, 1])
def test_split_k_does_not_depend_on_who_finishes_first(gpu, layout):
    """the 1024-frame contractions run two blocks per tile (gemm_bf16.h: gemm_bf16x3_splitk_floats); the second to finish adds
    the first one's partial sums to its own -- a + b = b + a, so forty launches give forty identical results"""
    import torch
    from tfkaldi_amd import _lib
    lib = gpu
    g = torch.Generator(device="cuda").manual_seed(5)
    M, N, K = 1024, 2048, 2048
    A = torch.randn(M, K, device="cuda", generator=g)
    B = torch.randn(*((N, K) if layout == 1 else (K, N)), device="cuda", generator=g)
    Ap, lda = _planes(lib, torch, A)
    Bp, ldb = _planes(lib, torch, B)
    outs = []
    for _ in range(40):
        C = torch.empty(M, N, device="cuda")
        _gemm(lib, torch, layout, Ap, lda, Bp, ldb, C, N, M, N, K)
        outs.append(C)
    torch.cuda.synchronize()
    for C in outs[1:]:
        assert torch.equal(C, outs[0])
    ref = A.double() @ (B.double().T if layout == 1 else B.double())
    assert float((outs[0].double() - ref).abs().max()) < 1e-3


def test_transpose_detecting(gpu):
    """A = I with an asymmetric B (integers up to 2^17: they need all three planes): a swapped row / column or a dropped
    plane cannot pass"""
    import torch
    n = 96
    A = torch.eye(n, device="cuda")
    B = (torch.arange(n, device="cuda")[:, None] * 1000 + torch.arange(n, device="cuda")[None, :]).float() + 0.5
    Ap, lda = _planes(gpu, torch, A)
    Bp, ldb = _planes(gpu, torch, B)
    C = torch.zeros(n, n, device="cuda")
    _gemm(gpu, torch, 0, Ap, lda, Bp, ldb, C, n, n, n, n)
    torch.cuda.synchronize()
    assert torch.equal(C, B)


@pytest.mark.timeout(1800)
def test_fp32_suites_in_the_emulated_arithmetic(gpu):
    """the fp32 engine suites against the float64 oracle at THEIR tolerances, with every engine the helpers build running
    compute_dtype = float32x3 (tests/util.py: TFK_TEST_DTYPE): every activation chain, multi-step training, the Adam known
    answer, evaluation / posteriors, layer-wise growth, k-engine data parallelism, tall micro-batches, stacked passes,
    BASELINE cfg2 at full size element-wise (ReLU and tanh)"""
    env = dict(os.environ, TFK_TEST_DTYPE="float32x3")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_engine_parity.py"),
                        os.path.join(ROOT, "tests", "test_gpu_stacked.py"), os.path.join(ROOT, "tests", "test_gpu_full_size.py"),
                        "-q", "-m", "gpu", "-k", "not optimiser_on_its_own and not bf16"], env=env, capture_output=True,
                       text=True, timeout=1700)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-1500:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_trainer_and_decoder_through_the_python_api(gpu, tmp_path):
    """DNN(compute_dtype="float32x3") through CrossEnthropyTrainer / Decoder on synthetic ark files, beside the exact-fp32
    engine from the same seed: losses within 2e-6 at the first step (round-off of two fp32 arithmetics) and 1e-3 after Adam
    has amplified it; posteriors within 1e-5"""
    from tfkaldi_amd import synthetic
    from tfkaldi_amd.neuralNetworks.classifiers import activation as act
    from tfkaldi_amd.neuralNetworks.classifiers.dnn import DNN
    from tfkaldi_amd.neuralNetworks.decoder import Decoder
    from tfkaldi_amd.neuralNetworks.trainer import CrossEnthropyTrainer
    from tfkaldi_amd.processing import batchdispenser, feature_reader, target_coder
    F_RAW, C, O = 8, 2, 12
    lengths = np.random.default_rng(0).integers(6, 30, size=30)
    paths = synthetic.write_corpus(str(tmp_path / "data"), 30, O, feat_dim=F_RAW, lengths=lengths, num_speakers=3)
    out = {}
    for dtype in ("float32", "float32x3"):
        reader = feature_reader.FeatureReader(paths["feats_scp"], paths["cmvn_scp"], paths["utt2spk"], C, 30)
        disp = batchdispenser.AlignmentBatchDispenser(reader, target_coder.AlignmentCoder(lambda x, y: x, O), 6, paths["alignments"])
        dnn = DNN(O, 2, 32, act.Dropout(act.TfActivation(act.Batchnorm(None), "relu"), 0.8), False, compute_dtype=dtype)
        tr = CrossEnthropyTrainer(dnn, F_RAW * (2 * C + 1), 30, 30, 1e-2, 1.0, 20, 2, seed=3)
        tr.initialize()
        losses = [tr.update_packed(disp.next_packed(tr.selector())) for _ in range(4)]
        prefix = str(tmp_path / dtype)
        tr.save_model(prefix)
        xs, _ = disp.get_batch()
        dec = Decoder(dnn, F_RAW * (2 * C + 1), 64)
        dec.restore(prefix)
        out[dtype] = (losses, dec(xs[0]))
        dec.close(); tr.close()
    l32, lx3 = out["float32"][0], out["float32x3"][0]
    assert abs(l32[0] - lx3[0]) <= 2e-6 * abs(l32[0]), (l32, lx3)
    assert np.allclose(l32, lx3, rtol=1e-3, atol=0), (l32, lx3)
    assert np.abs(out["float32"][1] - out["float32x3"][1]).max() <= 2e-3
