"""Replays the B=512 teacher-forced train step from its captured hipGraph a few times (for rocprofv3 --kernel-trace: the replayed
step has no host in it, so the trace shows the GPU-side concurrency of the step instead of the profiler's per-launch host cost).
python scripts/graph_replay.py [B] [replays]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from polyphonic_chord_texture_disentanglement_amd import functional as F_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.graph_step import GraphedTrainStep  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16')
opt = FusedClipAdam(m.parameters(), lr=1e-3)
x, c, pr = (torch.from_numpy(a).to(dev) for a in synth_batch(B, 1234))
m.use_philox(7, 0)
gs = GraphedTrainStep(m, opt, B)
for _ in range(n):
    gs(x, c, pr)
torch.cuda.synchronize()
F_.persist_check()
print('replayed', n)
