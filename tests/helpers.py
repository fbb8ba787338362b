"""Shared helpers for the parity tests (oracle side only -- never imported by the product)."""
import os
from collections import OrderedDict

import numpy as np
import torch

# Run-to-run noise of the weight gradients: every dW ends in fp32 atomicAdd of K-slab partials (wgrad.hip, gemm.hip split-K,
# texture.hip), so two runs of the same step differ in summation order.  Measured on MI355X (round 3; the probe script was retired in round 6, its result is kept:
# (profiles/r03_atomics_noise.jsonl; fp32 path, reduced B=6 and full geometry B=16, 6-8 repeats): <= 1.36e-6 of a tensor's
# max |g| per element (worst tensor: the conv bias, 49-way atomics), 2.4e-8 relative on the global norm, <= 3.8e-6 absolute on a
# loss.  ATOMICS_RTOL is that floor with a 7x margin; tests that compare two HIP runs of the same step (not HIP vs oracle) use it.
# bf16: reordered partial sums flip roundings of bf16-stored operands: 9.7e-4 of a tensor's max measured, ATOMICS_RTOL_BF16.
# That is the PTV_WGRAD_ORDERED=0 mode.  With the default ordered reductions (ptv_ordered_reductions) the floor is 0: two runs of a
# step give the same bits (tests/test_gpu_model.py::test_two_runs_of_a_training_trace_are_bit_identical).
ATOMICS_RTOL = 1e-5
ATOMICS_RTOL_BF16 = 5e-3
# One Adam step turns a gradient perturbation d into a parameter perturbation of up to lr * d / (|g| + eps): for |g| ~ eps = 1e-8 a
# 1e-10 reordering moves the parameter by ~lr * 1e-2.  Bound for parameters after Adam steps that started from noisy gradients:
ADAM_NOISE_FRAC_OF_LR = 2e-2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as f:
        return OrderedDict((k, f[k]) for k in f.files)


def reduced_params(requires_grad=False):
    sd = load_npz('reduced_state.npz')
    return OrderedDict((k, torch.from_numpy(v.copy()).requires_grad_(requires_grad))
                       for k, v in sd.items())


def full_shapes():
    f = load_npz('full_shapes.npz')
    return OrderedDict((str(n), tuple(int(t) for t in s.strip('()').split(',') if t.strip()))
                       for n, s in zip(f['names'], f['shapes']))


def full_params(seed=1234, requires_grad=False):
    from polyphonic_chord_texture_disentanglement_amd.synthetic import fill_state_dict
    sd = fill_state_dict(full_shapes(), seed)
    return OrderedDict((k, v.requires_grad_(requires_grad)) for k, v in sd.items())


class CoinList:
    """Replays a recorded coin-flip sequence (reference draw order)."""

    def __init__(self, coins):
        self.coins = list(coins)
        self.i = 0

    def __call__(self):
        v = self.coins[self.i]
        self.i += 1
        return v
