"""Which of the step's streams share a hardware queue?  (HIP folds all streams of a process onto GPU_MAX_HW_QUEUES = 4 queues; a queue
runs its dispatches in order.)  For every ordered pair (A, B) of {main, pool 0..3}: a 400-us kernel of 8 idle workgroups on A, then a
tiny kernel on B; B's completion time tells whether it had to wait for A's kernel.
    python scripts/micro/queue_map.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from polyphonic_chord_texture_disentanglement_amd import functional as F_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr  # noqa: E402

dev = torch.device('cuda:0')
torch.zeros(1, device=dev)
main = F_.cur_stream()
pool = [F_.Side(k).s for k in range(4)]
names = ['main'] + ['pool%d' % k for k in range(4)]
streams = [main] + pool
buf = torch.zeros(1024, device=dev)
out = torch.zeros(1024, device=dev)


def tiny(s):
    call('ptv_copy2d', ptr(out), 1024, ptr(buf), 1024, 1, 1024, 1.0, 0, s.cuda_stream)


for s in streams:
    call('ptv_debug_pin_cus', 8, 1024, 50, s.cuda_stream)
    tiny(s)
torch.cuda.synchronize()
print('rows: stream of the 400-us kernel; columns: stream of the tiny kernel; value: us until the tiny kernel finished')
print('%8s' % '' + ''.join('%8s' % n for n in names))
for i, a in enumerate(streams):
    row = []
    for j, b in enumerate(streams):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(b)
        call('ptv_debug_pin_cus', 8, 1024, 400, a.cuda_stream)
        tiny(b)
        e1.record(b)
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) * 1e3)
    print('%8s' % names[i] + ''.join('%8.0f' % v for v in row))
