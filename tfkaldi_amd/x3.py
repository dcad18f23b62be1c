"""Host-side view of the operand layout of `compute_dtype = float32x3` (csrc/x3_layout.h): an fp32 matrix [rows, ld] as three
bf16 planes p0 + p1 + p2 == x exactly, interleaved per 32 elements of the flat index i = row * ld + col -- element i of plane q at
(i // 32) * 96 + 32 q + i % 32.  Used by tests and tools that drive `tfk_split3` / `tfk_gemm_bf16x3` directly; the engine never
comes through here (its kernels write the planes themselves)."""
import ctypes

from . import _lib

BLOCK = 32


def padded_ld(cols, multiple=32):
    """leading dimension of a twin: a multiple of 8 is required, a multiple of 32 keeps rows on block boundaries"""
    return (cols + multiple - 1) // multiple * multiple


def split(lib, x, ld=None, stream=None):
    """the interleaved three-plane array of the CUDA float32 matrix x (tfk_split3): (bf16 tensor of 3 * rows * ld elements, ld)"""
    import torch
    rows, cols = x.shape
    ld = padded_ld(cols) if ld is None else ld
    n = (rows * ld + BLOCK - 1) // BLOCK * BLOCK
    out = torch.zeros(3 * n, dtype=torch.bfloat16, device=x.device)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream if stream is None else stream)
    _lib.check(lib.tfk_split3(st, ctypes.c_void_p(x.data_ptr()), x.stride(0), ctypes.c_void_p(out.data_ptr()), ld, rows, cols))
    return out, ld


def planes(arr, rows, ld):
    """the three planes of an interleaved array as [rows, ld] tensors (copies)"""
    n = rows * ld
    assert n % BLOCK == 0, "rows * ld must be a whole number of 32-element blocks to be viewed as planes"
    v = arr[:3 * n].view(n // BLOCK, 3, BLOCK)
    return [v[:, q, :].reshape(rows, ld) for q in range(3)]
