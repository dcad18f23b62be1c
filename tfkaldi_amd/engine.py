"""Python handle on one C-ABI engine (include/tfkaldi_hip.h): numpy in, numpy out.

This is plumbing only -- every FLOP of the hot path runs in libtfkaldi_hip.so.  The classes of
tfkaldi_amd.neuralNetworks (Trainer / Decoder / DNN) are built on it.
"""
from ctypes import byref, c_double, c_float, c_int, c_size_t, c_uint64, c_void_p

import numpy as np

from . import _lib
from ._lib import (ADAM_STEPS, BIASES, BN_BETA, BN_MOVING_MEAN, BN_MOVING_VAR, GLOBAL_STEP, INITIALISED_LAYERS,
                   LEARNING_RATE, LEARNING_RATE_FACT, SLOT_ADAM_M, SLOT_ADAM_V, SLOT_GRAD, SLOT_PARAM, WEIGHTS,
                   check)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Engine(object):
    """One engine per GPU.  `torch_state=True` keeps the persistent state in a torch CUDA tensor and
    runs on a torch stream so torch.distributed (RCCL) can all-reduce views of the gradient region."""

    def __init__(self, cfg, torch_state=False):
        self.lib = _lib.load()
        self.cfg = cfg
        self.F, self.L, self.H, self.O = cfg.input_dim, cfg.num_layers, cfg.num_units, cfg.output_dim
        self.batch_norm = bool(cfg.batch_norm)
        self._h = c_void_p()
        self._state = None
        self._stream = None
        self._cb = None
        self._layer_cb = None
        # Called before anything reads or writes the fp32 PARAMETERS from outside the optimiser (tensor get / set, the
        # layer-wise growth control ops).  The mixed-precision sharded exchange (dataparallel.BucketReducer) keeps the
        # fp32 masters of a span only on the rank that owns it; its hook refuses such an access until the masters have
        # been gathered collectively (DataParallel.gather_parameters) instead of handing out stale values.
        self.param_access_hook = None
        self.on_close = []
        self._raw_inflight = []   # (device tensor, event): inputs adopted in place that the engine may still be reading
        self._raw_pending = None
        if torch_state:
            import torch
            nbytes = c_size_t()
            check(self.lib.tfk_state_bytes(byref(cfg), byref(nbytes)))
            dev = torch.device("cuda", cfg.device)
            self._state = torch.zeros(nbytes.value // 4, dtype=torch.float32, device=dev)
            self._stream = torch.cuda.Stream(device=dev)
            torch.cuda.synchronize(dev)
            check(self.lib.tfk_create_ex(byref(cfg), c_void_p(self._state.data_ptr()), nbytes.value,
                                         c_void_p(self._stream.cuda_stream), byref(self._h)))
        else:
            check(self.lib.tfk_create(byref(cfg), byref(self._h)))

    # ---- lifetime ----
    def _raw_release(self, wait=False):
        """after a call that adopted a device tensor in place: mark the end of its use on the engine stream; drop the
        references whose work has completed (all of them with wait=True)"""
        if self._raw_pending is not None:
            import torch
            raw, ext = self._raw_pending
            ev = torch.cuda.Event()
            ev.record(ext)
            self._raw_inflight.append((raw, ev))
            self._raw_pending = None
        keep = []
        for raw, ev in self._raw_inflight:
            if wait:
                ev.synchronize()
            elif not ev.query():
                keep.append((raw, ev))
        self._raw_inflight = keep

    def close(self):
        if self._h:
            if self._raw_inflight or self._raw_pending:
                self._raw_release(wait=True)
            for fn in self.on_close:
                fn()
            del self.on_close[:]
            check(self.lib.tfk_destroy(self._h))
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- tensors / scalars ----
    def _shape(self, kind, layer):
        if kind == WEIGHTS:
            d_in = self.F if layer == 0 else self.H
            d_out = self.O if layer == self.L else self.H
            return (d_in, d_out)
        if kind == BIASES:
            return (self.O if layer == self.L else self.H,)
        return (self.H,)

    def _param_access(self, slot=SLOT_PARAM):
        if slot == SLOT_PARAM and self.param_access_hook is not None:
            self.param_access_hook()

    def get(self, kind, layer, slot=SLOT_PARAM):
        self._param_access(slot)
        out = np.empty(self._shape(kind, layer), dtype=np.float32)
        check(self.lib.tfk_tensor_get(self._h, kind, slot, layer, out.ctypes.data_as(c_void_p), out.size))
        return out

    def set(self, kind, layer, value, slot=SLOT_PARAM):
        v = _f32(value)
        if v.shape != self._shape(kind, layer):
            raise ValueError("tensor shape %s != %s" % (v.shape, self._shape(kind, layer)))
        self._param_access(slot)
        check(self.lib.tfk_tensor_set(self._h, kind, slot, layer, v.ctypes.data_as(c_void_p), v.size))

    def scalar(self, which):
        v = c_double()
        check(self.lib.tfk_scalar_get(self._h, which, byref(v)))
        return v.value

    def set_scalar(self, which, value):
        check(self.lib.tfk_scalar_set(self._h, which, float(value)))

    @property
    def global_step(self):
        return int(self.scalar(GLOBAL_STEP))

    def init_hidden_weights(self, rng):
        """neuralNetworks/classifiers/layer.py:39-44: hidden W ~ N(0, 1/sqrt(d_in)); everything else
        keeps the engine's defaults (biases 0, output layer 0: dnn.py:67-68)."""
        for l in range(self.L):
            d_in = self.F if l == 0 else self.H
            self.set(WEIGHTS, l, (rng.standard_normal((d_in, self.H)) / np.sqrt(d_in)).astype(np.float32))

    def model_tensors(self):
        """Everything the reference's model saver holds (dnn.py:129): W, b, BN beta + moving stats."""
        out = {}
        for l in range(self.L + 1):
            out["layer%d/weights" % l] = self.get(WEIGHTS, l)
            out["layer%d/biases" % l] = self.get(BIASES, l)
        if self.batch_norm:
            for l in range(self.L):
                out["layer%d/batch_norm/beta" % l] = self.get(BN_BETA, l)
                out["layer%d/batch_norm/moving_mean" % l] = self.get(BN_MOVING_MEAN, l)
                out["layer%d/batch_norm/moving_variance" % l] = self.get(BN_MOVING_VAR, l)
        out["initialisedlayers"] = np.array(int(self.scalar(INITIALISED_LAYERS)), dtype=np.int32)
        return out

    def load_model_tensors(self, tensors):
        for l in range(self.L + 1):
            self.set(WEIGHTS, l, tensors["layer%d/weights" % l])
            self.set(BIASES, l, tensors["layer%d/biases" % l])
        if self.batch_norm:
            for l in range(self.L):
                self.set(BN_BETA, l, tensors["layer%d/batch_norm/beta" % l])
                self.set(BN_MOVING_MEAN, l, tensors["layer%d/batch_norm/moving_mean" % l])
                self.set(BN_MOVING_VAR, l, tensors["layer%d/batch_norm/moving_variance" % l])
        if "initialisedlayers" in tensors:
            self.set_scalar(INITIALISED_LAYERS, int(tensors["initialisedlayers"]))

    # ---- training / evaluation ----
    @staticmethod
    def _host_batch(X, y):
        X = _f32(X)
        y = np.ascontiguousarray(y, dtype=np.int32)
        if X.ndim != 2 or y.ndim != 1 or X.shape[0] != y.shape[0]:
            raise ValueError("X %s / y %s are not [T, F] / [T]" % (X.shape, y.shape))
        return X, y

    def accumulate(self, X, y, last=False):
        X, y = self._host_batch(X, y)
        check(self.lib.tfk_accumulate(self._h, X.ctypes.data_as(c_void_p), X.shape[1], y.ctypes.data_as(c_void_p),
                                      X.shape[0], _lib.LAST_MICROBATCH if last else 0))

    def accumulate_device(self, x_ptr, ldx, y_ptr, T, last=False):
        """X / y already resident in HBM (raw device pointers)."""
        flags = _lib.DEVICE_PTRS | (_lib.LAST_MICROBATCH if last else 0)
        check(self.lib.tfk_accumulate(self._h, c_void_p(x_ptr), ldx, c_void_p(y_ptr), T, flags))

    # ---- device-side splice: unspliced frames in, spliced in HBM ----
    def _raw_device(self, raw, lens):
        """unspliced frames that already live in HBM (a float32 torch tensor, e.g. FeaturePlan.compute_device's result):
        (pointer, leading dimension, rows, utterance lengths); the engine's stream is ordered behind the tensor's producer"""
        import torch
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        if raw.dtype != torch.float32 or raw.dim() != 2 or raw.stride(1) != 1 or int(lens.sum()) != raw.shape[0]:
            raise ValueError("device frames must be a float32 [T, D] tensor with unit column stride whose rows match the "
                             "utterance lengths (sum %d)" % int(lens.sum()))
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(raw.device))
        stream = c_void_p()
        check(self.lib.tfk_stream(self._h, byref(stream)))
        ext = torch.cuda.ExternalStream(stream.value, device=raw.device)
        ext.wait_event(done)
        # Lifetime contract (round-2 advisor finding): the call returns while the splice kernel that reads `raw` is still
        # pending on the ENGINE's stream, which torch's caching allocator knows nothing about -- if the caller dropped the
        # tensor, its block could be handed out again and overwritten under the pending kernel.  The engine therefore keeps
        # a reference until an event recorded behind the consuming work has completed (_raw_release).  (Not
        # Tensor.record_stream: the allocator would later touch a stream that dies with the engine.)
        self._raw_pending = (raw, ext)
        return c_void_p(raw.data_ptr()), int(raw.stride(0)), int(raw.shape[0]), lens

    @staticmethod
    def _on_device(raw):
        return hasattr(raw, "is_cuda") and raw.is_cuda

    @staticmethod
    def _raw_batch(raw, lens):
        raw = _f32(raw)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        if raw.ndim != 2 or lens.ndim != 1 or int(lens.sum()) != raw.shape[0]:
            raise ValueError("raw %s / utterance lengths (sum %d) do not match" % (raw.shape, int(lens.sum())))
        return raw, lens

    @staticmethod
    def _cmvn_table(cmvn, raw, lens):
        """None or the [U, 2, D] (mean, std) table -> (keep-alive array, pointer)"""
        if cmvn is None:
            return None, c_void_p(None)
        cmvn = np.ascontiguousarray(cmvn, dtype=np.float32)
        if cmvn.shape != (lens.size, 2, raw.shape[1]):
            raise ValueError("cmvn table %s, expected %s" % (cmvn.shape, (lens.size, 2, raw.shape[1])))
        return cmvn, cmvn.ctypes.data_as(c_void_p)

    def accumulate_raw(self, raw, y, lens, context_width, last=False, cmvn=None):
        """`raw`: host [T, D] frames, or a float32 CUDA tensor (TFK_RAW_DEVICE)"""
        y = np.ascontiguousarray(y, dtype=np.int32)
        flags = _lib.LAST_MICROBATCH if last else 0
        if self._on_device(raw):
            ptr, ld, rows, lens = self._raw_device(raw, lens)
            flags |= _lib.RAW_DEVICE
            cols = raw.shape[1]
        else:
            raw, lens = self._raw_batch(raw, lens)
            ptr, ld, rows, cols = raw.ctypes.data_as(c_void_p), raw.shape[1], raw.shape[0], raw.shape[1]
        if cmvn is not None:
            cmvn = np.ascontiguousarray(cmvn, dtype=np.float32)
            if cmvn.shape != (lens.size, 2, cols):
                raise ValueError("cmvn table %s, expected %s" % (cmvn.shape, (lens.size, 2, cols)))
        cmvn_ptr = cmvn.ctypes.data_as(c_void_p) if cmvn is not None else c_void_p(None)
        if y.shape != (rows,):
            raise ValueError("targets %s do not match %d frames" % (y.shape, rows))
        try:
            check(self.lib.tfk_accumulate_raw(self._h, ptr, ld, y.ctypes.data_as(c_void_p), rows,
                                              lens.ctypes.data_as(c_void_p), lens.size, int(context_width), cmvn_ptr, flags))
        finally:
            self._raw_release()

    # ---- several micro-batches of one optimiser step in one call (stacked pass: include/tfkaldi_hip.h) ----
    def accumulate_stacked(self, X, y, seg_rows, last=False):
        """the micro-batches X[rows_0 + rows_1 + ..., F] back to back, seg_rows frames each: the same result as one
        accumulate per micro-batch, the GEMMs run once over all of them"""
        X, y = self._host_batch(X, y)
        seg = np.ascontiguousarray(seg_rows, dtype=np.int32)
        check(self.lib.tfk_accumulate_stacked(self._h, X.ctypes.data_as(c_void_p), X.shape[1], y.ctypes.data_as(c_void_p),
                                              X.shape[0], seg.ctypes.data_as(c_void_p), seg.size,
                                              _lib.LAST_MICROBATCH if last else 0))

    def accumulate_stacked_device(self, x_ptr, ldx, y_ptr, T, seg_rows, last=False):
        seg = np.ascontiguousarray(seg_rows, dtype=np.int32)
        flags = _lib.DEVICE_PTRS | (_lib.LAST_MICROBATCH if last else 0)
        check(self.lib.tfk_accumulate_stacked(self._h, c_void_p(x_ptr), ldx, c_void_p(y_ptr), T, seg.ctypes.data_as(c_void_p),
                                              seg.size, flags))

    def accumulate_stacked_raw(self, raw, y, lens, context_width, seg_utts, last=False, cmvn=None):
        """unspliced frames of all utterances back to back, seg_utts utterances per micro-batch (CMVN + splice on the device)"""
        raw, lens = self._raw_batch(raw, lens)
        y = np.ascontiguousarray(y, dtype=np.int32)
        seg = np.ascontiguousarray(seg_utts, dtype=np.int32)
        cmvn, cmvn_ptr = self._cmvn_table(cmvn, raw, lens)
        if y.shape != (raw.shape[0],):
            raise ValueError("targets %s do not match %d frames" % (y.shape, raw.shape[0]))
        check(self.lib.tfk_accumulate_stacked_raw(self._h, raw.ctypes.data_as(c_void_p), raw.shape[1], y.ctypes.data_as(c_void_p),
                                                  raw.shape[0], lens.ctypes.data_as(c_void_p), lens.size, int(context_width),
                                                  cmvn_ptr, seg.ctypes.data_as(c_void_p), seg.size,
                                                  _lib.LAST_MICROBATCH if last else 0))

    def eval_accumulate_stacked(self, X, y, seg_rows):
        """Trainer.evaluate's micro-batches X[rows_0 + rows_1 + ..., F] back to back in ONE call: rows are independent in
        evaluation mode, so the engine runs one pass of the GEMMs over all of them (tfk_eval_accumulate_stacked)"""
        X, y = self._host_batch(X, y)
        seg = np.ascontiguousarray(seg_rows, dtype=np.int32)
        check(self.lib.tfk_eval_accumulate_stacked(self._h, X.ctypes.data_as(c_void_p), X.shape[1], y.ctypes.data_as(c_void_p),
                                                   X.shape[0], seg.ctypes.data_as(c_void_p), seg.size, 0))

    def eval_accumulate_stacked_raw(self, raw, y, lens, context_width, seg_utts, cmvn=None):
        """the same from unspliced frames (CMVN + splice on the device), seg_utts utterances per micro-batch"""
        raw, lens = self._raw_batch(raw, lens)
        y = np.ascontiguousarray(y, dtype=np.int32)
        seg = np.ascontiguousarray(seg_utts, dtype=np.int32)
        cmvn, cmvn_ptr = self._cmvn_table(cmvn, raw, lens)
        if y.shape != (raw.shape[0],):
            raise ValueError("targets %s do not match %d frames" % (y.shape, raw.shape[0]))
        check(self.lib.tfk_eval_accumulate_stacked_raw(self._h, raw.ctypes.data_as(c_void_p), raw.shape[1],
                                                       y.ctypes.data_as(c_void_p), raw.shape[0], lens.ctypes.data_as(c_void_p),
                                                       lens.size, int(context_width), cmvn_ptr, seg.ctypes.data_as(c_void_p),
                                                       seg.size, 0))

    def eval_accumulate_raw(self, raw, y, lens, context_width, cmvn=None):
        raw, lens = self._raw_batch(raw, lens)
        y = np.ascontiguousarray(y, dtype=np.int32)
        cmvn, cmvn_ptr = self._cmvn_table(cmvn, raw, lens)
        check(self.lib.tfk_eval_accumulate_raw(self._h, raw.ctypes.data_as(c_void_p), raw.shape[1],
                                               y.ctypes.data_as(c_void_p), raw.shape[0],
                                               lens.ctypes.data_as(c_void_p), lens.size, int(context_width),
                                               cmvn_ptr, 0))

    def _out_buffer(self, rows):
        """[rows, O] result array; in pinned host memory (torch's caching pinned allocator) so that the engine can
        DMA straight into it -- the decode path moves 8 KB per frame back to the host"""
        if rows * self.O >= 1 << 16:
            try:
                import torch
                return torch.empty((rows, self.O), dtype=torch.float32, pin_memory=True).numpy()
            except Exception:  # noqa: BLE001  (no torch / no pinned memory: plain pageable memory works as well)
                pass
        return np.empty((rows, self.O), dtype=np.float32)

    def posteriors_raw(self, raw, lens, context_width, log_div_prior=False, raw_logits=False, cmvn=None):
        """`raw`: host [T, D] frames, or a float32 CUDA tensor (TFK_RAW_DEVICE: features that never left HBM)"""
        flags = (_lib.LOG_DIV_PRIOR if log_div_prior else 0) | (_lib.RAW_LOGITS if raw_logits else 0)
        if self._on_device(raw):
            ptr, ld, rows, lens = self._raw_device(raw, lens)
            flags |= _lib.RAW_DEVICE
            shape = (rows, raw.shape[1])
        else:
            raw, lens = self._raw_batch(raw, lens)
            ptr, ld, rows, shape = raw.ctypes.data_as(c_void_p), raw.shape[1], raw.shape[0], raw.shape
        out = self._out_buffer(rows)
        if cmvn is not None:
            cmvn = np.ascontiguousarray(cmvn, dtype=np.float32)
            if cmvn.shape != (lens.size, 2, shape[1]):
                raise ValueError("cmvn table %s, expected %s" % (cmvn.shape, (lens.size, 2, shape[1])))
        cmvn_ptr = cmvn.ctypes.data_as(c_void_p) if cmvn is not None else c_void_p(None)
        try:
            check(self.lib.tfk_posteriors_raw(self._h, ptr, ld, rows, lens.ctypes.data_as(c_void_p), lens.size,
                                              int(context_width), cmvn_ptr, out.ctypes.data_as(c_void_p), self.O, flags))
        finally:
            self._raw_release()
        return out

    # ---- CTC loss: frames [T, F] of U utterances + their label sequences ----
    def _ctc_args(self, X, utt_lens, labels, label_lens):
        X = _f32(X)
        utt_lens = np.ascontiguousarray(utt_lens, dtype=np.int32)
        label_lens = np.ascontiguousarray(label_lens, dtype=np.int32)
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        if int(utt_lens.sum()) != X.shape[0] or int(label_lens.sum()) != labels.size or utt_lens.size != label_lens.size:
            raise ValueError("CTC micro-batch: frames %s / utterance lengths / labels do not match" % (X.shape,))
        return X, utt_lens, labels, label_lens

    def accumulate_ctc_raw(self, raw, utt_lens, context_width, labels, label_lens, last=False, cmvn=None, train=True):
        """CTC micro-batch from UNSPLICED frames: CMVN + splice on the device (as accumulate_raw)"""
        raw, utt_lens, labels, label_lens = self._ctc_args(raw, utt_lens, labels, label_lens)
        cmvn, cmvn_ptr = self._cmvn_table(cmvn, raw, utt_lens)
        fn = self.lib.tfk_accumulate_ctc_raw if train else self.lib.tfk_eval_accumulate_ctc_raw
        check(fn(self._h, raw.ctypes.data_as(c_void_p), raw.shape[1], raw.shape[0], utt_lens.ctypes.data_as(c_void_p),
                 utt_lens.size, int(context_width), cmvn_ptr, labels.ctypes.data_as(c_void_p),
                 label_lens.ctypes.data_as(c_void_p), _lib.LAST_MICROBATCH if (last and train) else 0))

    def accumulate_ctc(self, X, utt_lens, labels, label_lens, last=False):
        X, utt_lens, labels, label_lens = self._ctc_args(X, utt_lens, labels, label_lens)
        check(self.lib.tfk_accumulate_ctc(self._h, X.ctypes.data_as(c_void_p), X.shape[1], X.shape[0],
                                          utt_lens.ctypes.data_as(c_void_p), utt_lens.size,
                                          labels.ctypes.data_as(c_void_p), label_lens.ctypes.data_as(c_void_p),
                                          _lib.LAST_MICROBATCH if last else 0))

    def eval_accumulate_ctc(self, X, utt_lens, labels, label_lens):
        X, utt_lens, labels, label_lens = self._ctc_args(X, utt_lens, labels, label_lens)
        check(self.lib.tfk_eval_accumulate_ctc(self._h, X.ctypes.data_as(c_void_p), X.shape[1], X.shape[0],
                                               utt_lens.ctypes.data_as(c_void_p), utt_lens.size,
                                               labels.ctypes.data_as(c_void_p), label_lens.ctypes.data_as(c_void_p),
                                               0))

    # ---- the optimiser step in parts (data parallel: Adam per reduced span, see dataparallel.BucketReducer) ----
    def apply_begin(self):
        check(self.lib.tfk_apply_begin(self._h))

    def apply_span(self, offset_floats, num_floats):
        check(self.lib.tfk_apply_span(self._h, int(offset_floats), int(num_floats)))

    def apply_end(self):
        loss = c_float()
        check(self.lib.tfk_apply_end(self._h, byref(loss)))
        return float(loss.value)

    def apply(self):
        loss = c_float()
        check(self.lib.tfk_apply(self._h, byref(loss)))
        return float(loss.value)

    def apply_enqueue(self):
        """the optimiser step on the streams, not waited for: apply_end() collects the loss"""
        check(self.lib.tfk_apply_enqueue(self._h))

    def eval_accumulate(self, X, y):
        X, y = self._host_batch(X, y)
        check(self.lib.tfk_eval_accumulate(self._h, X.ctypes.data_as(c_void_p), X.shape[1],
                                           y.ctypes.data_as(c_void_p), X.shape[0], 0))

    def eval_finish(self):
        loss = c_float()
        check(self.lib.tfk_eval_finish(self._h, byref(loss)))
        return float(loss.value)

    def halve_learning_rate(self):
        check(self.lib.tfk_halve_learning_rate(self._h))

    def add_layer(self):
        self._param_access()  # (a control op on the parameters: refused while the fp32 masters are sharded)
        check(self.lib.tfk_add_layer(self._h))

    def init_last_layer(self):
        self._param_access()
        check(self.lib.tfk_init_last_layer(self._h))

    # ---- decoding ----
    def set_prior(self, prior):
        p = _f32(prior)
        check(self.lib.tfk_set_prior(self._h, p.ctypes.data_as(c_void_p), p.size))

    def posteriors(self, X, log_div_prior=False, raw_logits=False):
        X = _f32(X)
        out = self._out_buffer(X.shape[0])
        flags = (_lib.LOG_DIV_PRIOR if log_div_prior else 0) | (_lib.RAW_LOGITS if raw_logits else 0)
        check(self.lib.tfk_posteriors(self._h, X.ctypes.data_as(c_void_p), X.shape[1], X.shape[0],
                                      out.ctypes.data_as(c_void_p), self.O, flags))
        return out

    # ---- data parallelism ----
    def reduce_region(self):
        ptr, n = c_void_p(), c_size_t()
        check(self.lib.tfk_reduce_region(self._h, byref(ptr), byref(n)))
        return ptr.value, n.value

    def bucket_order(self):
        """the order in which accumulate(last=True) announces the buckets: scalars + BN increments, the weight
        matrices from the output layer down, the bias / beta gradients (include/tfkaldi_hip.h)"""
        L = self.L
        return [L + 2] + list(range(L + 1)) + [L + 1]

    def buckets(self):
        nb = c_int()
        check(self.lib.tfk_num_buckets(self._h, byref(nb)))
        out = []
        for b in range(nb.value):
            off, n = c_size_t(), c_size_t()
            check(self.lib.tfk_reduce_bucket(self._h, b, byref(off), byref(n)))
            out.append((off.value, n.value))
        return out

    def reduce_view(self):
        """torch view of the reduce region (requires torch_state=True)."""
        if self._state is None:
            raise RuntimeError("reduce_view needs Engine(..., torch_state=True)")
        ptr, n = self.reduce_region()
        off = (ptr - self._state.data_ptr()) // 4
        return self._state[off:off + n]

    def param_view(self):
        """torch view of the parameter arena [P] (same offsets as the gradient part of the reduce region); the
        sharded exchange step all-gathers the updated parameters straight into it (requires torch_state=True)"""
        if self._state is None:
            raise RuntimeError("param_view needs Engine(..., torch_state=True)")
        ptr, _ = self.reduce_region()
        return self._state[0:(ptr - self._state.data_ptr()) // 4]

    def params_touched(self):
        check(self.lib.tfk_params_touched(self._h))

    def twins_from_params(self, offset, n, stream=None):
        """the fp32 parameters [offset, offset + n) (whole weight matrices) were written behind the optimiser's back: rebuild
        what the contractions read from them (emulated fp32: the three-plane twins) on `stream` (None: the engine's).  True
        when that is now current for the span, False when nothing was done (the caller then uses params_touched)"""
        cur = c_int()
        check(self.lib.tfk_twins_from_params(self._h, int(offset), int(n), c_void_p(stream), byref(cur)))
        return bool(cur.value)

    def shadow_view(self):
        """torch bfloat16 view of the arena-mirroring weight shadow (mixed precision, every leading dimension a multiple
        of 8; requires torch_state=True), element offsets = those of the fp32 weight arena; None when there is none"""
        if self._state is None:
            raise RuntimeError("shadow_view needs Engine(..., torch_state=True)")
        import torch
        ptr, n, mirrors = c_void_p(), c_size_t(), c_int()
        check(self.lib.tfk_shadow_region(self._h, byref(ptr), byref(n), byref(mirrors)))
        if not mirrors.value or not n.value:
            return None
        off = (ptr.value - self._state.data_ptr()) // 4
        return self._state[off:off + (n.value + 1) // 2].view(torch.bfloat16)[:n.value]

    def apply_writes_shadow(self):
        """between apply_begin and apply_end: does apply_span write the bf16 shadow with the update?"""
        d = c_int()
        check(self.lib.tfk_apply_writes_shadow(self._h, byref(d)))
        return bool(d.value)

    def param_checksum(self, which=0):
        """64-bit integer checksum of the fp32 parameter arena (0) or the bf16 shadow (1); synchronises"""
        v = c_uint64()
        check(self.lib.tfk_param_checksum(self._h, int(which), byref(v)))
        return int(v.value)

    def set_bucket_callback(self, fn):
        """fn(bucket) is called from accumulate(last=True) once the bucket's kernels are enqueued."""
        if fn is None:
            self._cb = None
            check(self.lib.tfk_set_bucket_callback(self._h, _lib.BUCKET_FN(), None))
            return
        self._cb = _lib.BUCKET_FN(lambda user, bucket: fn(bucket))
        check(self.lib.tfk_set_bucket_callback(self._h, self._cb, None))

    def set_layer_callback(self, fn):
        """fn(layer) is called right before kernels that read the parameters of `layer` (0 .. L; -1 = all) are
        enqueued (tfk_set_layer_callback): the hook of the asynchronous parameter all-gather."""
        if fn is None:
            self._layer_cb = None
            check(self.lib.tfk_set_layer_callback(self._h, _lib.BUCKET_FN(), None))
            return
        self._layer_cb = _lib.BUCKET_FN(lambda user, layer: fn(layer))
        check(self.lib.tfk_set_layer_callback(self._h, self._layer_cb, None))

    def zero_accumulators(self):
        check(self.lib.tfk_zero_accumulators(self._h))

    def set_later_microbatches(self, later):
        check(self.lib.tfk_set_later_microbatches(self._h, int(later)))

    @property
    def torch_stream(self):
        return self._stream

    # ---- misc ----
    def synchronize(self):
        check(self.lib.tfk_synchronize(self._h))

    def profile_begin(self):
        check(self.lib.tfk_profile_begin(self._h))

    def profile_end(self):
        stats = (_lib.TfkKernelStat * 32)()
        n = c_int()
        check(self.lib.tfk_profile_end(self._h, stats, 32, byref(n)))
        return [dict(name=stats[i].name.decode(), launches=int(stats[i].launches), total_ms=stats[i].total_ms,
                     flops=stats[i].flops, bytes=stats[i].bytes) for i in range(min(n.value, 32))]

    def debug_fetch(self, what, layer, T):
        cols = self.O if what == _lib.DBG_LOGITS else self.H
        out = np.empty((T, cols), dtype=np.float32)
        check(self.lib.tfk_debug_fetch(self._h, what, layer, out.ctypes.data_as(c_void_p), out.size))
        return out
