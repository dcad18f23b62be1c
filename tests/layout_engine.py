"""Test double with the REAL engine's state layout and protocol, and arithmetic that is exact in float32.

`LayoutEngine` restates compute_layout() of tfkaldi_amd/csrc/engine.hip (W-first arena, 64-float aligned spans, the bias /
beta vectors behind the weight matrices, the 64-float scalar block and the BN increments behind the gradient arena; buckets
as tfk_reduce_bucket numbers them; announcement order of tfk_accumulate(TFK_LAST_MICROBATCH)) so that the product's
DataParallel / BucketReducer run on CPU (gloo) against the bucket sizes of a BASELINE configuration -- cfg2: 26 M parameters,
seven weight spans of 3.6 / 16.8 / 16.4 MB -- with 2, 4 or 8 ranks.  The layout restatement is pinned to the library by
tests/test_dataparallel_layout.py::test_layout_matches_the_library (tfk_state_bytes needs no GPU).

There is no network behind it: the "gradient" of micro-batch `mb` at arena offset i is a hash of (i, mb) on a 2^-11 grid, the BN
statistics live on a 2^-3 grid and the BN decay is 1/2, so every sum over micro-batches is EXACT in float32 whatever the order --
a sharded run must therefore equal the serial run bit for bit, Adam included (it is element-wise).  What is being tested is the
exchange machinery: coalescing, the n % (4 * world) shard rule, idle ranks, layer-wise growth, the asynchronous gathers and who
waits for them (every "forward pass" digests the parameters of layer l right after announcing the read, as the real engine's
kernels would read them), the bf16-shadow variant and the fp32 masters that stay sharded under it.
"""
import numpy as np
import torch


def up(x, a):
    return (x + a - 1) // a * a


class Layout(object):
    def __init__(self, F, L, H, O, batch_norm=True):
        self.F, self.L, self.H, self.O, self.bn = F, L, H, O, batch_norm
        ldH, ldO = up(H, 4), up(O, 4)
        self.ldH = ldH
        off = 0
        self.w, self.b, self.beta = [], [], []
        for l in range(L + 1):
            d_in = F if l == 0 else H
            ld_out = ldO if l == L else ldH
            n = up(d_in * ld_out, 64)
            self.w.append((off, n))
            off += n
        self.vec_off = off
        for l in range(L + 1):
            n = up(ldO if l == L else ldH, 64)
            self.b.append((off, n))
            off += n
        for l in range(L + 1):
            n = up(ldH, 64) if (batch_norm and l < L) else 0
            self.beta.append((off, n))
            off += n
        self.P = off
        self.E = up(2 * L * ldH, 64) if batch_norm else 0
        self.reduce_floats = self.P + 64 + self.E
        self.mirrors = all((ldO if l == L else ldH) % 8 == 0 for l in range(L + 1))

    def state_bytes(self, bf16=False):
        shadow = up(self.vec_off, 128) // 2 if (bf16 and self.mirrors) else 0
        return (4 * self.P + 64 + 2 * self.E + shadow) * 4

    def buckets(self):
        L = self.L
        out = [self.w[L - b] for b in range(L + 1)]
        out.append((self.vec_off, self.P - self.vec_off))
        out.append((self.P, 64 + self.E))
        return out


def grad_values(off, n, mb):
    """exact-grid pseudo-gradient of micro-batch `mb` for arena elements [off, off + n): multiples of 2^-11 in [-4, 4)"""
    i = np.arange(off, off + n, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(mb + 1) * np.uint64(40503)) >> np.uint64(7)
    return ((h & np.uint64(0x3FFF)).astype(np.float32) - 8192.0) / 2048.0


def stat_values(n, mb, which):
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(97) + np.uint64(mb + 3) * np.uint64(31) + np.uint64(which) * np.uint64(7)) & np.uint64(15)
    return (h.astype(np.float32) - 8.0) / 8.0  # multiples of 2^-3 in [-1, 1)


class LayoutEngine(object):
    torch_stream = None
    BN_DECAY = 0.5

    def __init__(self, layout, bf16=False, nact=None, lr=1e-3):
        self.lay, self.bf16 = layout, bf16
        self.L = layout.L
        self.nact = layout.L if nact is None else nact  # active hidden layers (layer-wise growth)
        P = layout.P
        self.params = torch.zeros(P, dtype=torch.float32)
        # deterministic non-trivial initial parameters
        self.params.numpy()[:] = grad_values(0, P, 12345) / 4.0
        self.region = torch.zeros(layout.reduce_floats, dtype=torch.float32)
        self.m = np.zeros(P, dtype=np.float32)
        self.v = np.zeros(P, dtype=np.float32)
        self.mov = np.zeros(layout.E, dtype=np.float32)
        # (as the real engine: the shadow exists but is NOT current until the first forward pass or optimiser step)
        self.shadow = torch.zeros(layout.vec_off, dtype=torch.bfloat16) if (bf16 and layout.mirrors) else None
        self._shadow_dirty = self.shadow is not None
        self.later = 0
        self.cb = self.layer_cb = None
        self.param_access_hook = None
        self.on_close = []
        self.fresh = True
        self.apply_open = False
        self.lr, self.t, self.global_step = lr, 0, 0
        # what the forward passes "read": (optimiser steps done so far, layer) -> checksum of the span as it was at that
        # moment (parameters only change in apply, so every pass between two applies must see the same value)
        self.digests = {}
        self.closed = False

    # ---- layout / views ----
    def buckets(self):
        return self.lay.buckets()

    def bucket_order(self):
        L = self.L
        return [L + 2] + list(range(L + 1)) + [L + 1]

    def reduce_view(self):
        return self.region

    def param_view(self):
        return self.params

    def shadow_view(self):
        return self.shadow

    def params_touched(self):
        if self.shadow is not None:
            self._shadow_dirty = True

    def set_bucket_callback(self, fn):
        self.cb = fn

    def set_layer_callback(self, fn):
        self.layer_cb = fn

    def set_later_microbatches(self, later):
        self.later = later

    def zero_accumulators(self):
        self.region.zero_()
        self.fresh = False

    def close(self):
        for fn in self.on_close:
            fn()
        del self.on_close[:]
        self.closed = True

    # ---- "forward": announce the read of each layer, then digest what is there ----
    def _read_layers(self):
        if self._shadow_dirty:
            if self.layer_cb:
                self.layer_cb(-1)
            self.shadow[:] = self.params[:self.lay.vec_off].to(torch.bfloat16)
            self._shadow_dirty = False
        for l in range(self.L + 1):
            if self.layer_cb:
                self.layer_cb(l)
            off, n = self.lay.w[l]
            if self.shadow is not None:
                words = self.shadow[off:off + n].view(torch.int16).numpy().astype(np.int64)
            else:
                words = self.params[off:off + n].numpy().view(np.int32).astype(np.int64)
            # (biases / beta are read with the layer as well)
            vo, vn = self.lay.b[l]
            vec = self.params[vo:vo + vn].numpy().view(np.int32).astype(np.int64)
            d = int(words.sum()) ^ int(vec.sum())
            assert self.digests.setdefault((self.global_step, l), d) == d, "parameters changed between two applies"

    def accumulate(self, mb, frames, last=False):
        """micro-batch `mb` (an integer id) of `frames` frames"""
        self._read_layers()
        r = self.region.numpy()
        lay, P = self.lay, self.lay.P
        if self.fresh:
            r[:] = 0
            self.fresh = False
        for l in range(self.L + 1):
            if l < self.L and l >= self.nact:
                continue  # above the active depth: zero gradient
            for off, n in (lay.w[l], lay.b[l], lay.beta[l]):
                if n:
                    r[off:off + n] += grad_values(off, n, mb)
        r[P] += 0.5 * mb + 1.0
        r[P + 1] += frames
        r[P + 2] += 1
        if lay.E:
            d = self.BN_DECAY
            e = r[P + 64:P + 64 + lay.E]
            e[:] = d * e + (1 - d) * stat_values(lay.E, mb, 0)
        if last:
            if lay.E and self.later > 0:
                r[P + 64:] *= np.float32(self.BN_DECAY ** self.later)
            if self.cb:
                for b in self.bucket_order():
                    self.cb(b)

    # ---- optimiser protocol ----
    def apply_begin(self):
        assert not self.apply_open
        r = self.region.numpy()
        P = self.lay.P
        if self.fresh:
            r[:] = 0
        self._loss, self._frames, nmb = float(r[P]), float(r[P + 1]), float(r[P + 2])
        if self.lay.E:
            self.mov[:] = np.float32(self.BN_DECAY ** nmb) * self.mov + r[P + 64:]
        self.t += 1
        b1, b2 = 0.9, 0.999
        self._lr_t = np.float32(self.lr * np.sqrt(1 - b2 ** self.t) / (1 - b1 ** self.t))
        if self.shadow is not None and self._shadow_dirty:  # (tfk_apply_begin: an arena-mirroring shadow is made current)
            if self.layer_cb:
                self.layer_cb(-1)
            self.shadow[:] = self.params[:self.lay.vec_off].to(torch.bfloat16)
            self._shadow_dirty = False
        self._direct = self.shadow is not None
        self.apply_open = True

    def apply_writes_shadow(self):
        assert self.apply_open
        return self._direct

    def apply_span(self, off, n):
        assert self.apply_open
        P = self.lay.P
        if off >= P or n == 0:
            return
        n = min(n, P - off)
        sl = slice(off, off + n)
        b1, b2, eps = np.float32(0.9), np.float32(0.999), np.float32(1e-8)
        g = np.clip(self.region.numpy()[sl] / np.float32(self._frames), -1.0, 1.0).astype(np.float32)
        self.m[sl] = b1 * self.m[sl] + (np.float32(1) - b1) * g
        self.v[sl] = b2 * self.v[sl] + (np.float32(1) - b2) * (g * g)
        w = self.params.numpy()
        w[sl] = w[sl] - (self._lr_t * self.m[sl]) / (np.sqrt(self.v[sl]) + eps)
        if self._direct and off < self.lay.vec_off:
            hi = min(off + n, self.lay.vec_off)
            self.shadow[off:hi] = self.params[off:hi].to(torch.bfloat16)

    def apply_end(self):
        assert self.apply_open
        self.apply_open = False
        self.fresh = True
        self.global_step += 1
        return self._loss / self._frames

    def apply(self):
        self.apply_begin()
        self.apply_span(0, self.lay.P)
        return self.apply_end()

    def param_checksum(self, which=0):
        if which == 1:
            w = self.shadow.view(torch.int16).numpy().astype(np.int64)
        elif which == 2:
            w = self.params[self.lay.vec_off:].numpy().view(np.int32).astype(np.int64)
        else:
            w = self.params.numpy().view(np.int32).astype(np.int64)
        return int((w * ((np.arange(w.size) & 0xffff) + 1)).sum()) & 0xFFFFFFFFFFFFFFFF

    # ---- evaluation: only the scalar tail ----
    def eval_accumulate(self, mb, frames):
        self._read_layers()
        r = self.region.numpy()
        P = self.lay.P
        if self.fresh:
            r[:] = 0
            self.fresh = False
        r[P] += 0.25 * mb + 2.0
        r[P + 1] += frames

    def eval_finish(self):
        r = self.region.numpy()
        P = self.lay.P
        out = float(r[P]) / float(r[P + 1])
        self.fresh = True
        return out

    def get_params(self):
        """what a checkpoint would read: refused while the fp32 masters are sharded"""
        if self.param_access_hook is not None:
            self.param_access_hook()
        return self.params.numpy().copy()
