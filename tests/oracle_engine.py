"""Test double: the float64 oracle behind the Engine methods that tfkaldi_amd.dataparallel uses, with the
same reduce-region layout idea ([G | loss, frames, #micro-batches, pad | BN moving-average increments]).
Lets the world_size-2 gloo tests drive the PRODUCT's DataParallel class on CPU."""
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

    def eval_accumulate(self, X, y):
        self.o.eval_accumulate(X, y)
        r = self.region.numpy()
        r[self.P] = self.o.batch_loss
        r[self.P + 1] = self.o.num_frames

    def eval_finish(self):
        r = self.region.numpy()
        self.o.batch_loss, self.o.num_frames = float(r[self.P]), int(round(r[self.P + 1]))
        r[:] = 0
        return self.o.eval_finish()
