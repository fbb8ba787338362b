"""Batch adapter of the reference (`dataset_loaders.py`): `MusicDataLoaders` and `TrainingVAE`.

The POP909 files the reference's `dataset.py` needs are not available, so `get_loaders` serves the
synthetic generator of `synthetic.py` in the reference's batch layout
`(mel_segments, prs, pr_mats, p_grids, chord, dt_x)` (`dataset.py:117-118`); `_batch_to_inputs` applies
the reference's casts (`dataset_loaders.py:28-34`) and returns the THREE tensors the model consumes
(the reference returns four and cannot run)."""
import torch

from ._lib import call, ptr, stream_ptr
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


def batch_transform(pr, chord14, shift=None, index=None, check=False):
    """The reference's per-item transform (dataset.py:88-112 over converter.py:65-164) for a whole batch on the device:
    pr uint8 [N,32,128] (2 onset / 1 sustain / 0 silence), chord14 f32 [N,8,14], shift int32 [B] semitones, index int32 [B]
    items of the batch (None = the first B) -> (pr_mat f32 [B,32,128], x int64 [B,32,16,6], c f32 [B,8,36]).
    check=True synchronises and raises IndexError where the reference would (more than 14 onsets in a step)."""
    assert pr.is_cuda and pr.dtype == torch.uint8 and pr.is_contiguous(), 'pr: contiguous cuda uint8 [N,32,128]'
    chord14 = chord14.float().contiguous()
    B = int(index.numel() if index is not None else (shift.numel() if shift is not None else pr.shape[0]))
    dev = pr.device
    pr_mat = torch.empty(B, 32, 128, device=dev, dtype=torch.float32)
    x = torch.empty(B, 32, 16, 6, device=dev, dtype=torch.int64)
    c = torch.empty(B, 8, 36, device=dev, dtype=torch.float32)
    err = torch.zeros(1, device=dev, dtype=torch.int32)
    call('ptv_batch_transform', ptr(pr), ptr(chord14), ptr(index.int().contiguous() if index is not None else None),
         ptr(shift.int().contiguous() if shift is not None else None), ptr(pr_mat), ptr(x), ptr(c), ptr(err), B, stream_ptr())
    if check and int(err.item()):
        raise IndexError('a time step holds more than 14 simultaneous onsets (converter.py:141 raises here)')
    return pr_mat, x, c


class DeviceBatcher:
    """The training set resident in HBM (uint8 piano-rolls: 4 KB per 2-bar item) served as ready model inputs without
    touching the host: one epoch enumerates every (item, shift) pair of ArrangementDataset (dataset.py:63-69: ids
    0 .. N*(shift_high-shift_low+1)-1, item = id // n_shift, shift = id % n_shift + shift_low) in a device-side
    permutation (the DataLoader's shuffle=True of dataset.py:279) and runs ptv_batch_transform per batch.  Yields the
    reference's 6-tuple batch layout (mel_segments, prs, pr_mats, p_grids, chord, dt_x) with the unused slots empty."""

    def __init__(self, pr, chord14, batch_size, shift_low=-6, shift_high=5, shuffle=True, seed=3345, drop_last=False, device=None):
        dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.pr = torch.as_tensor(pr).to(dev, torch.uint8).contiguous()
        self.chord = torch.as_tensor(chord14).to(dev, torch.float32).contiguous()
        self.batch_size, self.shift_low, self.n_shift = batch_size, shift_low, shift_high - shift_low + 1
        self.shuffle, self.drop_last = shuffle, drop_last
        self.gen = torch.Generator(device=dev).manual_seed(seed)
        self.n = self.pr.shape[0] * self.n_shift

    def __len__(self):
        return self.n // self.batch_size if self.drop_last else (self.n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        dev = self.pr.device
        ids = torch.randperm(self.n, device=dev, generator=self.gen) if self.shuffle else torch.arange(self.n, device=dev)
        empty = torch.empty(0, device=dev)
        for i in range(len(self)):
            b = ids[i * self.batch_size:(i + 1) * self.batch_size]
            index = torch.div(b, self.n_shift, rounding_mode='floor').int()
            shift = (b % self.n_shift + self.shift_low).int()
            pr_mat, x, c = batch_transform(self.pr, self.chord, shift, index)
            yield empty, empty, pr_mat, x, c, empty


class MusicDataLoaders(DataLoaders):

    @staticmethod
    def get_loaders(seed, bs_train, bs_val, portion=8, shift_low=-6, shift_high=5, num_bar=2, contain_chord=True,
                    random_train=True, random_val=False, n_train_batch=8, n_val_batch=2, device_bank=None):
        """device_bank = (pr uint8 [N,32,128], chord14 [N,8,14]) serves the reference's augmented epochs from HBM
        (DeviceBatcher); otherwise the synthetic three-tensor generator."""
        if device_bank is not None:
            pr, chord = device_bank
            n_val = max(1, pr.shape[0] // (portion + 1))                 # dataset.py:241-245,273-276: 1/(portion+1) validates, unshifted
            train = DeviceBatcher(pr[:-n_val], chord[:-n_val], bs_train, shift_low, shift_high, random_train, seed)
            val = DeviceBatcher(pr[-n_val:], chord[-n_val:], bs_val, 0, 0, random_val, seed + 1)
            return MusicDataLoaders(train, val, bs_train, bs_val)
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
