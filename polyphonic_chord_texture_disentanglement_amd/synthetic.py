"""Synthetic 2-bar piano-roll batches.

There is no POP909 data in this build (reference `dataset.py:242-281` needs files that
are absent), so every test / bench / golden fixture uses this generator.  It emits the
three tensors the training step consumes, in the layouts the reference's data pipeline
defines:

* ``pr_mat [B,32,128] f32`` -- duration (in 16th steps) at onset cells, else 0
  (`converter.py:87-113`, "piano_roll_to_target").
* ``x [B,32,16,6] int64`` -- the PianoTree grid of `converter.py:116-147` called with the
  arguments of `dataset.py:98-104` (max_note_count=16, pad=130, sos=128, eos=129,
  dur_pad=2): row 0 is ``<sos>``, then the notes of the step in ascending pitch as
  ``[pitch, 5-bit MSB-first binary of (dur-1)]``, then ``<eos>``, then ``<pad>`` rows.
* ``c [B,8,36] f32`` -- one-hot root(12) | binary chroma(12) | one-hot bass(12)
  (`converter.py:150-164`).

Pure numpy; deterministic for a given (B, seed).
"""
import numpy as np

PITCH_SOS, PITCH_EOS, PITCH_PAD, DUR_PAD = 128, 129, 130, 2
MAX_SIMU_NOTE, NUM_STEP = 16, 32


def pr_mat_to_grid(pr_mat):
    """[32,128] duration matrix -> [32,16,6] int64 PianoTree grid (layout above)."""
    grid = np.full((NUM_STEP, MAX_SIMU_NOTE, 6), DUR_PAD, dtype=np.int64)
    grid[:, :, 0] = PITCH_PAD
    grid[:, 0, 0] = PITCH_SOS
    for t in range(NUM_STEP):
        pitches = np.nonzero(pr_mat[t])[0]          # ascending
        assert len(pitches) <= MAX_SIMU_NOTE - 2
        for k, p in enumerate(pitches):
            d = int(pr_mat[t, p]) - 1
            grid[t, k + 1, 0] = p
            grid[t, k + 1, 1:] = [(d >> s) & 1 for s in (4, 3, 2, 1, 0)]
        grid[t, len(pitches) + 1, 0] = PITCH_EOS
    return grid


def synth_batch(B, seed):
    """Returns (x int64 [B,32,16,6], c f32 [B,8,36], pr_mat f32 [B,32,128]) numpy arrays."""
    rng = np.random.RandomState(seed)
    pr_mat = np.zeros((B, NUM_STEP, 128), dtype=np.float32)
    x = np.zeros((B, NUM_STEP, MAX_SIMU_NOTE, 6), dtype=np.int64)
    c = np.zeros((B, 8, 36), dtype=np.float32)
    for b in range(B):
        for t in range(NUM_STEP):
            if rng.rand() < 0.5:
                k = rng.randint(1, 7)
                ps = rng.choice(np.arange(36, 96), k, replace=False)
                for p in ps:
                    pr_mat[b, t, p] = rng.randint(1, min(NUM_STEP - t, 16) + 1)
        x[b] = pr_mat_to_grid(pr_mat[b])
        for i in range(8):
            root = rng.randint(12)
            chroma = (rng.rand(12) < 0.3).astype(np.float32)
            bass = rng.randint(12)
            c[b, i, root] = 1.0
            c[b, i, 12:24] = chroma
            c[b, i, 24 + bass] = 1.0
    return x, c, pr_mat


def synth_raw_bank(N, seed):
    """A bank of N raw 2-bar items in the layout the reference's dataset holds BEFORE its per-item transform
    (dataset.py:88-112): accompaniment piano-rolls `pr` uint8 [N,32,128] with 2 at onsets and 1 on the sustained cells
    (converter.py:35-47, later notes overwrite earlier ones) and raw chords `chord14` f32 [N,8,14] = [root, 12 chroma
    bits, bass] (the operand of converter.py:150-164).  Input of the device batch transform (csrc/data.hip)."""
    rng = np.random.RandomState(seed)
    pr = np.zeros((N, NUM_STEP, 128), dtype=np.uint8)
    chord = np.zeros((N, 8, 14), dtype=np.float32)
    for b in range(N):
        for t in range(NUM_STEP):
            if rng.rand() < 0.5:
                k = rng.randint(1, 7)
                ps = rng.choice(np.arange(36, 96), k, replace=False)
                for p in ps:
                    d = rng.randint(1, min(NUM_STEP - t, 16) + 1)
                    pr[b, t, p] = 2
                    pr[b, t + 1: t + d, p] = 1
        chord[b, :, 0] = rng.randint(0, 12, 8)
        chord[b, :, 1:13] = rng.rand(8, 12) < 0.3
        chord[b, :, 13] = rng.randint(0, 12, 8)
    return pr, chord


def fill_state_dict(shapes, seed):
    """Deterministic filler weights for parity fixtures.

    `shapes` is an ordered mapping name -> shape (the 81 `state_dict` entries).  Tensor i is
    drawn from its own CPU `torch.Generator(seed + i)`: matrices / conv kernels U(-a, a) with
    a = 1/sqrt(fan_in), bias vectors U(-0.1, 0.1), the three learned start tokens U(0, 1)
    (the reference initialises those with `torch.rand`, ptvae.py:42,256-259).  The same call
    fills the reference (in `tests/golden/make_golden.py`) and this build (in the tests).
    """
    import torch
    from collections import OrderedDict
    out = OrderedDict()
    for i, (name, shape) in enumerate(shapes.items()):
        g = torch.Generator().manual_seed(int(seed) + i)
        u = torch.rand(tuple(shape), generator=g, dtype=torch.float32)
        if name.endswith(('dec_init_input', 'dur_sos_token', 'init_input')):
            t = u
        elif len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= int(s)
            t = (2.0 * u - 1.0) / float(np.sqrt(fan_in))
        else:
            t = (2.0 * u - 1.0) * 0.1
        out[name] = t
    return out
