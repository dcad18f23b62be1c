"""CPU BASELINE (test/bench infrastructure, NOT product code): the optimiser step of the reference's
CrossEnthropyTrainer restated with PyTorch CPU ops in fp32 on all host cores.

The reference's own CPU path is TensorFlow 0.x, which is not installable here; this
restatement of the identical work -- forward + backward of every micro-batch (trainer.py:310-332), gradient
sum, / num_frames, clip to [-1, 1], TF-formulation Adam (:174-184), BN moving-average update -- is the
labelled stand-in timed by bench.py's `cpu_baseline` leg (kind "port").  It is cross-checked against
oracle/dnn_oracle.py in tests/test_oracle.py.
"""
import math

import torch


class TorchCpuTrainer(object):
    def __init__(self, input_dim, num_layers, num_units, output_dim, nonlin="relu", batch_norm=True,
                 init_learning_rate=1e-3, bn_decay=0.999, bn_epsilon=1e-3, threads=None):
        if threads:
            torch.set_num_threads(threads)
        self.L, self.bn, self.nonlin = num_layers, batch_norm, nonlin
        self.lr, self.bn_decay, self.bn_eps = init_learning_rate, bn_decay, bn_epsilon
        dims = [input_dim] + [num_units] * num_layers + [output_dim]
        self.W = [torch.zeros(dims[l], dims[l + 1], requires_grad=True) for l in range(num_layers + 1)]
        self.b = [torch.zeros(dims[l + 1], requires_grad=True) for l in range(num_layers + 1)]
        self.beta = [torch.zeros(num_units, requires_grad=True) for _ in range(num_layers)] if batch_norm else []
        self.mov_mean = [torch.zeros(num_units) for _ in range(num_layers)]
        self.mov_var = [torch.ones(num_units) for _ in range(num_layers)]
        self.params = self.W + self.b + self.beta
        self.G = [torch.zeros_like(p) for p in self.params]
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.loss, self.frames, self.t = 0.0, 0, 0

    def set_hidden_weights(self, weights):
        with torch.no_grad():
            for l, w in enumerate(weights):
                self.W[l].copy_(torch.as_tensor(w))

    def accumulate(self, X, y):
        act = {"relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh, "linear": lambda t: t}[self.nonlin]
        a = torch.as_tensor(X)
        for l in range(self.L):
            z = a @ self.W[l] + self.b[l]
            if self.bn:
                mu, var = z.mean(0), z.var(0, unbiased=False)
                with torch.no_grad():
                    self.mov_mean[l].mul_(self.bn_decay).add_((1 - self.bn_decay) * mu)
                    self.mov_var[l].mul_(self.bn_decay).add_((1 - self.bn_decay) * var)
                z = (z - mu) * torch.rsqrt(var + self.bn_eps) + self.beta[l]
            a = act(z)
        logits = a @ self.W[self.L] + self.b[self.L]
        loss = torch.nn.functional.cross_entropy(logits, torch.as_tensor(y).long(), reduction="sum")
        grads = torch.autograd.grad(loss, self.params)
        with torch.no_grad():
            for G, g in zip(self.G, grads):
                G.add_(g)
        self.loss += float(loss.detach())
        self.frames += int(a.shape[0])

    def apply(self):
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - 0.999 ** self.t) / (1 - 0.9 ** self.t)
        with torch.no_grad():
            for p, G, m, v in zip(self.params, self.G, self.m, self.v):
                g = (G / float(self.frames)).clamp_(-1.0, 1.0)
                m.mul_(0.9).add_(0.1 * g)
                v.mul_(0.999).add_(0.001 * g * g)
                p.sub_(lr_t * m / (v.sqrt() + 1e-8))
                G.zero_()
        avg = self.loss / self.frames
        self.loss, self.frames = 0.0, 0
        return avg
