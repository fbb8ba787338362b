"""Parity of the benched dtype against the reference-generated golden (tests/golden/full_tf1_b16.npz: full init_model() geometry,
B = 16, teacher-forced, filler weights): max |d loss| over the 11 losses, relative error of the global gradient norm, worst
per-tensor gradient-norm error.  Modes: fp32, bf16 (the benched path), bf16 with fp32 storage of the saved tensors
(functional.BF16_STORAGE = False: bf16 MFMA operands only).  Usage: python scripts/bf16_parity.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from polyphonic_chord_texture_disentanglement_amd import functional as F_  # noqa: E402
from bench import golden_parity  # noqa: E402

if __name__ == '__main__':
    out = {}
    for mode in sys.argv[1:] or ['fp32', 'bf16', 'bf16_fp32_storage']:
        if mode == 'bf16_fp32_storage':
            F_.BF16_STORAGE = False
        out[mode] = golden_parity('fp32' if mode == 'fp32' else 'bf16', torch.device('cuda:0'), detail=True)
        F_.BF16_STORAGE = True
        print(mode, json.dumps(out[mode]), flush=True)
