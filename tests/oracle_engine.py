"""Test double: the float64 oracle behind the Engine methods that tfkaldi_amd.dataparallel uses, with the
same reduce-region layout idea ([G | loss, frames, #micro-batches, pad | BN moving-average increments]) and the
same optimiser protocol (apply_begin / apply_span on a span of the flat parameter arena / apply_end, param_view).
Lets the world_size-2 gloo tests drive the PRODUCT's DataParallel class -- both exchange steps -- on CPU."""
import numpy as np
import torch


class OracleEngine(object):
    torch_stream = None

    def __init__(self, oracle):
        self.o = oracle
        self.keys = []
        for l in range(oracle.L, -1, -1):  # bucket b = layer L - b
            ks = ["W%d" % l, "b%d" % l] + (["beta%d" % l] if oracle.bn and l < oracle.L else [])
            self.keys.append(ks)
        self._buckets, off = [], 0
        for ks in self.keys:
            n = sum(oracle.params()[k].size for k in ks)
            n = (n + 15) // 16 * 16  # spans are padded like the engine's (there to 64 floats): they divide by 4 * world
            self._buckets.append((off, n))
            off += n
        self.P = off
        self.E = 2 * oracle.L * oracle.H if oracle.bn else 0
        self._buckets.append((off, 4 + self.E))
        self.region = torch.zeros(off + 4 + self.E, dtype=torch.float64)
        self.later = 0
        self.cb = None
        self._mov0 = None
        self._nmb = 0

    def buckets(self):
        return list(self._buckets)

    def reduce_view(self):
        return self.region

    def zero_accumulators(self):
        self.region.zero_()

    def set_later_microbatches(self, later):
        self.later = later

    def set_bucket_callback(self, fn):
        self.cb = fn

    def set_layer_callback(self, fn):
        """as Engine.set_layer_callback: fn(layer) before parameters are read (the asynchronous all-gather's hook)"""
        self.layer_cb = fn

    def sync_params(self):
        """parameters may still be arriving (all-gather in flight): wait, then adopt the flat arena"""
        if getattr(self, "layer_cb", None):
            self.layer_cb(-1)
        if hasattr(self, "params_flat"):
            self._unflat(self.params_flat.numpy(), self.o.params())

    def _snapshot(self):
        if self._mov0 is None and self.o.bn:
            self._mov0 = ([m.copy() for m in self.o.mov_mean], [v.copy() for v in self.o.mov_var])

    def _pack(self):
        o = self.o
        r = self.region.numpy()
        r[:] = 0
        for (off, _), ks in zip(self._buckets, self.keys):
            for k in ks:
                r[off:off + o.G[k].size] = o.G[k].ravel()
                off += o.G[k].size
        r[self.P:self.P + 3] = [o.batch_loss, o.num_frames, self._nmb]
        if o.bn:
            d = o.bn_decay
            w = d ** self.later
            e = []
            for l in range(o.L):
                m0 = self._mov0[0][l] if self._mov0 else o.mov_mean[l]
                v0 = self._mov0[1][l] if self._mov0 else o.mov_var[l]
                e.append((o.mov_mean[l] - d ** self._nmb * m0) * w)
                e.append((o.mov_var[l] - d ** self._nmb * v0) * w)
            r[self.P + 4:] = np.concatenate(e)

    def accumulate(self, X, y, last=False):
        self.sync_params()
        self._snapshot()
        self.o.accumulate(X, y)
        self._nmb += 1
        if last:
            self._pack()
            if self.cb:
                for b in range(len(self._buckets)):
                    self.cb(b)

    def apply(self):
        o = self.o
        r = self.region.numpy()
        if self._nmb == 0:
            self._snapshot()
        for (off, _), ks in zip(self._buckets, self.keys):
            for k in ks:
                o.G[k] = r[off:off + o.G[k].size].reshape(o.G[k].shape).copy()
                off += o.G[k].size
        o.batch_loss, o.num_frames = float(r[self.P]), int(round(r[self.P + 1]))
        total_mb = int(round(r[self.P + 2]))
        if o.bn:
            d = o.bn_decay
            e = r[self.P + 4:].reshape(2 * o.L, o.H)
            for l in range(o.L):
                o.mov_mean[l] = d ** total_mb * self._mov0[0][l] + e[2 * l]
                o.mov_var[l] = d ** total_mb * self._mov0[1][l] + e[2 * l + 1]
        self._mov0, self._nmb = None, 0
        r[:] = 0
        return o.apply()

    # ---- the optimiser in parts, on spans of the flat arena (include/tfkaldi_hip.h: tfk_apply_begin / _span / _end) ----
    def _flat(self, d):
        out = np.zeros(self.P)
        for (off, _), ks in zip(self._buckets, self.keys):
            for k in ks:
                out[off:off + d[k].size] = d[k].ravel()
                off += d[k].size
        return out

    def _unflat(self, flat, d):
        for (off, _), ks in zip(self._buckets, self.keys):
            for k in ks:
                d[k][...] = flat[off:off + d[k].size].reshape(d[k].shape)
                off += d[k].size

    def param_view(self):
        if not hasattr(self, "params_flat"):
            self.params_flat = torch.from_numpy(self._flat(self.o.params()))
        return self.params_flat

    def apply_begin(self):
        o = self.o
        r = self.region.numpy()
        if self._nmb == 0:
            self._snapshot()
        self._frames = float(r[self.P + 1])
        self._loss = float(r[self.P])
        total_mb = int(round(r[self.P + 2]))
        if o.bn:
            d = o.bn_decay
            e = r[self.P + 4:].reshape(2 * o.L, o.H)
            for l in range(o.L):
                o.mov_mean[l] = d ** total_mb * self._mov0[0][l] + e[2 * l]
                o.mov_var[l] = d ** total_mb * self._mov0[1][l] + e[2 * l + 1]
        lr = o.learning_rate()
        o.adam_t += 1
        self._lr_t = lr * np.sqrt(1.0 - o.b2 ** o.adam_t) / (1.0 - o.b1 ** o.adam_t)
        self.sync_params()
        self.param_view().numpy()[:] = self._flat(o.params())
        self._m, self._v = self._flat(o.m), self._flat(o.v)

    def apply_span(self, off, n):
        o = self.o
        n = min(n, self.P - off)
        if n <= 0:
            return
        sl = slice(off, off + n)
        g = np.clip(self.region.numpy()[sl] / self._frames, -1.0, 1.0)
        self._m[sl] = o.b1 * self._m[sl] + (1 - o.b1) * g
        self._v[sl] = o.b2 * self._v[sl] + (1 - o.b2) * g * g
        self.params_flat.numpy()[sl] -= self._lr_t * self._m[sl] / (np.sqrt(self._v[sl]) + o.eps)

    def apply_end(self):
        o = self.o
        # Adam moments of spans this rank did not update stay at their old values in a sharded step: a real rank
        # never reads them again for those spans either (it always updates the same sub-spans), but the test double
        # shares one oracle state, so only the parameters are compared across ranks
        self._unflat(self.params_flat.numpy(), o.params())
        self._unflat(self._m, o.m)
        self._unflat(self._v, o.v)
        o.global_step += 1
        o._zero_accumulators()
        self._mov0, self._nmb = None, 0
        self.region.numpy()[:] = 0
        return self._loss / self._frames

    def eval_accumulate(self, X, y):
        self.sync_params()
        self.o.eval_accumulate(X, y)
        r = self.region.numpy()
        r[self.P] = self.o.batch_loss
        r[self.P + 1] = self.o.num_frames

    def eval_finish(self):
        r = self.region.numpy()
        self.o.batch_loss, self.o.num_frames = float(r[self.P]), int(round(r[self.P + 1]))
        r[:] = 0
        return self.o.eval_finish()
