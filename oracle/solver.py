"""Optimiser + learning-rate policies of the reference, restated in numpy.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/__init__.py).

Follows /root/reference/core/solver.py:
  _get_model_learning_rate :138-202 (period_step / custom_step / poly / plateau, slow start)
  _get_model_optimizer     :204-219 (Adam beta1 .9 beta2 .99; Momentum .9)
  plateau_decay            :246-254
and the TF-1.13 formulas they call:
  tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; theta -= lr_t*m/(sqrt(v)+eps)
  tf.train.exponential_decay(staircase=True), piecewise_constant, polynomial_decay (cycle=False)
  tf.train.MomentumOptimizer: accum = mom*accum + g; theta -= lr*accum (nesterov: lr*(g + mom*accum))
"""
import math

import numpy as np


def learning_rate(policy, global_step, base_lr=1e-3, decay_step=100000, decay_rate=0.1,
                  boundaries=None, values=None, total_steps=1000, end_lr=1e-6, power=0.9,
                  plateau_lr=None, slow_start_step=0, slow_start_lr=1e-4):
    if policy == "period_step":
        lr = base_lr * decay_rate ** math.floor(global_step / decay_step)
    elif policy == "custom_step":
        # tf.train.piecewise_constant: values[0] when x <= boundaries[0], values[i] when b[i-1] < x <= b[i]
        lr = values[-1]
        for b, v in zip(boundaries, values):
            if global_step <= b:
                lr = v
                break
    elif policy == "poly":
        gs = min(global_step, total_steps)
        lr = (base_lr - end_lr) * (1 - gs / total_steps) ** power + end_lr
    elif policy == "plateau":
        lr = base_lr if plateau_lr is None else plateau_lr
    else:
        raise ValueError("Not supported learning policy.")
    if slow_start_step > 0 and global_step < slow_start_step:
        lr = slow_start_lr
    return lr


def plateau_update(lr, factor, min_lr):
    """solver.py:251: lr <- max(lr * factor, min_lr)."""
    return max(lr * factor, min_lr)


class TFAdam(object):
    def __init__(self, beta1=0.9, beta2=0.99, eps=1e-8):
        self.b1, self.b2, self.eps = beta1, beta2, eps
        self.t = 0
        self.m, self.v = {}, {}

    def step(self, params, grads, lr):
        """params/grads: dict name -> float ndarray (updated in place)."""
        self.t += 1
        lr_t = lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for k, g in grads.items():
            g = np.asarray(g, dtype=params[k].dtype)
            if k not in self.m:
                self.m[k] = np.zeros_like(params[k])
                self.v[k] = np.zeros_like(params[k])
            self.m[k] += (1 - self.b1) * (g - self.m[k])
            self.v[k] += (1 - self.b2) * (g * g - self.v[k])
            params[k] -= lr_t * self.m[k] / (np.sqrt(self.v[k]) + self.eps)


class TFMomentum(object):
    def __init__(self, momentum=0.9, use_nesterov=False):
        self.mom, self.nesterov = momentum, use_nesterov
        self.acc = {}

    def step(self, params, grads, lr):
        for k, g in grads.items():
            g = np.asarray(g, dtype=params[k].dtype)
            if k not in self.acc:
                self.acc[k] = np.zeros_like(params[k])
            self.acc[k] = self.mom * self.acc[k] + g
            if self.nesterov:
                params[k] -= lr * (g + self.mom * self.acc[k])
            else:
                params[k] -= lr * self.acc[k]
