"""Batch adapter of the reference (`dataset_loaders.py`): `MusicDataLoaders` and `TrainingVAE`.

The POP909 files the reference's `dataset.py` needs are not available, so `get_loaders` serves the
synthetic generator of `synthetic.py` in the reference's batch layout
`(mel_segments, prs, pr_mats, p_grids, chord, dt_x)` (`dataset.py:117-118`); `_batch_to_inputs` applies
the reference's casts (`dataset_loaders.py:28-34`) and returns the THREE tensors the model consumes
(the reference returns four and cannot run)."""
import torch

from .amc_dl.torch_plus import DataLoaders, TrainingInterface
from .synthetic import synth_batch

SEED = 3345            # dataset.py:13


class _SyntheticLoader:
    def __init__(self, n_batch, batch_size, seed):
        self.n_batch, self.batch_size, self.seed = n_batch, batch_size, seed

    def __len__(self):
        return self.n_batch

    def __iter__(self):
        for i in range(self.n_batch):
            x, c, pr = synth_batch(self.batch_size, self.seed + i)
            zeros = torch.zeros(self.batch_size, 1)
            yield zeros, zeros, torch.from_numpy(pr), torch.from_numpy(x), torch.from_numpy(c), zeros


class MusicDataLoaders(DataLoaders):

    @staticmethod
    def get_loaders(seed, bs_train, bs_val, portion=8, shift_low=-6, shift_high=5, num_bar=2, contain_chord=True,
                    random_train=True, random_val=False, n_train_batch=8, n_val_batch=2):
        train = _SyntheticLoader(n_train_batch, bs_train, seed)
        val = _SyntheticLoader(n_val_batch, bs_val, seed + 10 ** 6)
        return MusicDataLoaders(train, val, bs_train, bs_val)

    def batch_to_inputs(self, batch):
        _, _, pr_mat, x, c, _ = batch
        return x.to(self.device).long(), c.to(self.device).float(), pr_mat.to(self.device).float()


class TrainingVAE(TrainingInterface):

    def _batch_to_inputs(self, batch):
        _, _, pr_mat, x, c, _ = batch
        pr_mat = pr_mat.to(self.device).float()
        x = x.to(self.device).long()
        c = c.to(self.device).float()
        return x, c, pr_mat
