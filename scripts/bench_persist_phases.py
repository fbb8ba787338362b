import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts')
import bench_persist as bp
from polyphonic_chord_texture_disentanglement_amd import functional as F_
from polyphonic_chord_texture_disentanglement_amd._lib import lib
dev = bp.dev; bf = bp.bf
M, H, T = 512, 1024, 32
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
hall = torch.zeros(T + 1, M, H, device=dev); hall[0] = rn(M, H) * 0.5
c = dict(gi=(rn(T, M, 3 * H) * 0.5).to(bf), gi_step=M * 3 * H, gi_ld=3 * H, gi2=(rn(M, 3 * H) * 0.5).to(bf), gi2_step=0, gi2_ld=3 * H,
         w16=(rn(3 * H, H) / H ** 0.5).to(bf), b_hh=rn(3 * H) * 0.1, hall=hall, hall16=torch.zeros(T + 1, M, H, device=dev, dtype=bf),
         gates=torch.rand(T, 4, M, H, device=dev).to(bf), lengths=None, reverse=False)
for name, fl in (('all on', 0), ('no wait', 1), ('no publish', 2), ('no wait, no publish', 3), ('no product', 4), ('no product, wait, publish', 7),
                 ('no stores at all (8)', 8), ('nothing but the cell (15)', 15), ('all on', 0)):
    lib().ptv_gru_persist_load_policy(100 + fl)
    t = bp.timeit(lambda: F_.gru_persist_fwd(M, H, T, [c]))
    print('%-32s %7.1f us  %5.2f us/step' % (name, t, t / T), flush=True)
lib().ptv_gru_persist_load_policy(100)
