"""Host-side view of the operand layout of `compute_dtype = float32x3` (csrc/x3_layout.h): an fp32 matrix [rows, ld] (ld a
multiple of 32) as three bf16 planes p0 + p1 + p2 == x exactly, TILED in units of 2 rows x 32 columns -- 384 bytes, one 128-byte
line per plane holding the unit's two rows:

    element (row, col), plane q  at  ((row // 2) * (ld // 32) + col // 32) * 192 + 64 q + 32 (row % 2) + col % 32

Used by tests and tools that drive `tfk_split3` / `tfk_gemm_bf16x3` directly; the engine never comes through here (its kernels
write the planes themselves)."""
import ctypes

from . import _lib

UNIT_COLS = 32


def padded_ld(cols):
    """leading dimension of a twin: whole 32-column units"""
    return (cols + UNIT_COLS - 1) // UNIT_COLS * UNIT_COLS


def elems(rows, ld):
    """bf16 elements the twin of an [rows, ld] matrix occupies (rows rounded up to even)"""
    return (rows + 1) // 2 * (ld // UNIT_COLS) * 192


def split(lib, x, ld=None, stream=None):
    """the tiled three-plane twin of the CUDA float32 matrix x (tfk_split3): (bf16 tensor of elems(rows, ld) elements, ld)"""
    import torch
    rows, cols = x.shape
    ld = padded_ld(cols) if ld is None else ld
    out = torch.zeros(elems(rows, ld), dtype=torch.bfloat16, device=x.device)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream if stream is None else stream)
    _lib.check(lib.tfk_split3(st, ctypes.c_void_p(x.data_ptr()), x.stride(0), ctypes.c_void_p(out.data_ptr()), ld, rows, cols))
    return out, ld


def planes(arr, rows, ld):
    """the three planes of a twin as [rows, ld] tensors (copies; an odd last row is cut off the padding)"""
    r2 = (rows + 1) // 2
    v = arr[:elems(rows, ld)].view(r2, ld // UNIT_COLS, 3, 2, UNIT_COLS)  # [row pair, unit, plane, row parity, column]
    return [v[:, :, q].permute(0, 2, 1, 3).reshape(2 * r2, ld)[:rows] for q in range(3)]
