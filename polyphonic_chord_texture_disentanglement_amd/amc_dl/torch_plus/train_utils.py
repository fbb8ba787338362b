"""Helpers of the reference's amc_dl/torch_plus/train_utils.py (same names and semantics)."""
import numpy as np
import torch


def epoch_time(start_time, end_time):
    """train_utils.py:6-10 -> (minutes, seconds)"""
    elapsed = end_time - start_time
    mins = int(elapsed / 60)
    return mins, int(elapsed - mins * 60)


def join_fn(*items, ext='pt'):
    """train_utils.py:13-14: join_fn('a','b',ext='pt') -> 'a_b.pt'"""
    return '_'.join(items) + '.' + ext


def _inv_sigmoid_ramp(i):
    # 1 / (1 + e^{10 (i - 1/2)}): a ramp meant for i in [0, 1] -- the trainer feeds it the integer
    # batch counter, so it saturates after two steps; reproduced as is.
    return 1 / (1 + np.exp(10 * (i - 0.5)))


def scheduled_sampling(i, high=0.7, low=0.05):
    """train_utils.py:17-21"""
    return (high - low) * _inv_sigmoid_ramp(i) + low


def kl_anealing(i, high=0.1, low=0.):
    """train_utils.py:24-30"""
    hh, ll = 1 - low, 1 - high
    return 1 - ((hh - ll) * _inv_sigmoid_ramp(i) + ll)


def get_zs_from_dists(dists, sample=False):
    """train_utils.py:33-34"""
    return [d.rsample() if sample else d.mean for d in dists]


def standard_normal(shape, device=None):
    """train_utils.py:37-42; the prior follows `device` instead of being forced onto .cuda()."""
    from ...ptvae import HipNormal
    return HipNormal(torch.zeros(shape, device=device), torch.ones(shape, device=device))


def kl_with_normal(dist):
    """train_utils.py:45-49: MEAN over every element of KL(dist || N(0, 1)), on the HIP kernels."""
    from ... import functional as F_
    return F_.KlFn.apply(dist.mean, dist.scale)
