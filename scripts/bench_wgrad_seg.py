"""ptv_wgrad_batch with K segments, standalone: the notes GRU's three weight-gradient products of the B = 512 step (K = 15 x 16384 rows, 7 live
note steps, the live prefix of every step from the synthetic batch) -- clipped (seg_n) against unclipped (seg_n = R everywhere: same slab plan)"""
import sys
import torch
sys.path.insert(0, '.')
from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr, wgrad_batch
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch

dev = torch.device('cuda:0')
B = 512; R = 32 * B; T = 15; K = T * R
x = torch.from_numpy(synth_batch(B, 99)[0]).to(dev)
row_live = torch.zeros(R, device=dev, dtype=torch.int32)
pt = torch.empty(B * 480, device=dev, dtype=torch.int32); dt = torch.empty(B * 2400, device=dev, dtype=torch.int32)
cnt = torch.zeros(3, device=dev, dtype=torch.int32)
call('ptv_pianotree_targets_rows', ptr(x), B, 1, ptr(pt), ptr(dt), ptr(cnt), ptr(row_live), stream_ptr())
perm = torch.empty(R, device=dev, dtype=torch.int32)
call('ptv_rows_by_length', ptr(row_live), ptr(perm), R, 15, stream_ptr())
len_s = torch.empty(R, device=dev, dtype=torch.int32)
call('ptv_gather_rows', ptr(len_s), ptr(row_live), ptr(perm), R, 1, 0, 0, 1, stream_ptr())
seg = torch.empty(T, device=dev, dtype=torch.int32)
call('ptv_rows_seg_counts', ptr(len_s), R, T, ptr(seg), stream_ptr())
full = torch.full((T,), R, device=dev, dtype=torch.int32)
top = cnt[2:3].clone()
print('live note steps', int(top) + 1, 'seg_n', seg.tolist(), 'live pairs %.3f of the live steps' % (float(seg.sum()) / ((int(top) + 1) * R)))
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(1)
dgi = torch.randn(K, 1536, device=dev, generator=g).to(bf); dgh = torch.randn(K, 512, device=dev, generator=g).to(bf)
HN = torch.randn(K, 512, device=dev, generator=g).to(bf); TOK = torch.randn(K, 128, device=dev, generator=g)
Cs = [torch.zeros(1024, 512, device=dev), torch.zeros(512, 512, device=dev), torch.zeros(1536, 128, device=dev)]
bs = [torch.zeros(1024, device=dev), torch.zeros(512, device=dev)]


SLABS = 0


def jobs(sg):
    c = dict(K=K, k_top=top, k_unit=R, slabs=SLABS)
    if sg is not None:
        c.update(seg_n=sg, seg_unit=R, seg_period=T)
    return [dict(c, M=1024, N=512, A=dgi[:, :1024], B=HN, C=Cs[0], colsum_a=bs[0]), dict(c, M=512, N=512, A=dgh, B=HN, C=Cs[1], colsum_a=bs[1]),
            dict(c, M=1536, N=128, A=dgi, B=TOK, C=Cs[2])]


for name, sg in (('no segments (round-5 plan)', None), ('segments = R (same plan, nothing clipped)', full), ('segments clipped', seg),
                 ('no segments (round-5 plan)', None), ('segments clipped', seg)):
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); wgrad_batch(jobs(sg)); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print('%-46s %8.1f us' % (name, best * 1e3))

for SLABS in (16, 30, 60, 120, 240):
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); wgrad_batch(jobs(seg)); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print('segments clipped, slabs wanted = %3d            %8.1f us' % (SLABS, best * 1e3))

# the note-summary bi-GRU's products of one direction (K = 16 positions x 16384 rows, 8 live): the default slab plan against finer ones
K2 = 16 * R
dgi2 = torch.randn(K2, 384, device=dev, generator=g).to(bf); dgh2 = torch.randn(K2, 384, device=dev, generator=g).to(bf)
x2 = torch.randn(K2, 128, device=dev, generator=g); h2 = torch.randn(K2, 128, device=dev, generator=g).to(bf)
C2 = [torch.zeros(384, 128, device=dev), torch.zeros(384, 128, device=dev)]; b2 = [torch.zeros(384, device=dev), torch.zeros(384, device=dev)]
top2 = torch.tensor([7], device=dev, dtype=torch.int32)
for SL in (0, 16, 32, 64, 128, 256):
    js = [dict(M=384, N=128, K=K2, A=dgi2, B=x2, C=C2[0], colsum_a=b2[0], k_top=top2, k_unit=R, slabs=SL),
          dict(M=384, N=128, K=K2, A=dgh2, B=h2, C=C2[1], colsum_a=b2[1], k_top=top2, k_unit=R, slabs=SL)]
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); wgrad_batch(js); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print('note-summary products (384 x 128 x 262144, 8 of 16 positions live), slabs wanted = %3d   %8.1f us' % (SL, best * 1e3))
