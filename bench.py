"""bench.py -- 2-bar piano-roll samples/sec of the full polyphonic-VAE train step on MI355X.

    python bench.py [--gpus N --steps K --warmup W --batch 512 --precision bf16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
    python bench.py --gpus N          (no launcher: bench.py starts the N ranks itself, one fresh process per GPU)

A step = zero_grad -> model('train', x, c, pr_mat, tfr=1, beta, weights) -> backward -> RCCL
all-reduce (N>1) -> fused global-norm clip + Adam -> MinExponentialLR.step, on synthetic batches
resident in HBM (BASELINE.json configs[1]: batch 512 per GPU, bf16 MFMA GRU/Linear products,
z_dim 256+256, teacher-forced decoder).  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# forward FLOP/sample with the loop-invariant notes-GRU input product hoisted (SURVEY.md §8d,
# Appendix C); a train step is 3x forward.  Used for the whole-step TFLOP/s figure only.
GFLOP_PER_SAMPLE_TRAIN = 6.15


def cpu_baseline(batch=16, timed=3):
    """The CPU oracle (a restatement of the reference's as-written algorithm, validated against the reference in
    tests/test_oracle_vs_golden.py) timed on this host's cores: full train steps (fwd + bwd + clip + Adam) at BASELINE
    configs[0] (B = 16), teacher-forced and free-running, 1 warm-up + `timed` steps each, min and median.
    The reference itself, timed in the build container on 8 vCPU: profiles/r02_reference_cpu_timing.json."""
    import statistics
    from oracle.ptvae_oracle import Oracle, clip_and_adam_step
    from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE
    from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))          # the oracle's matmuls are small: more threads only add overhead
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    ref = DisentangleVAE.init_model(torch.device('cpu'))
    x, c, pr = (torch.from_numpy(a) for a in synth_batch(batch, 1234))
    out = {}
    for tfr in (1.0, 0.0):
        params = {k: v.detach().clone().requires_grad_(True) for k, v in ref.state_dict().items()}
        plist = list(params.values())
        m = [torch.zeros_like(p) for p in plist]
        v = [torch.zeros_like(p) for p in plist]
        gen = torch.Generator().manual_seed(7)
        times = []
        for step in range(1 + timed):
            eps = [torch.randn(batch, 256, generator=gen) for _ in range(2)]
            t0 = time.perf_counter()
            for p in plist:
                p.grad = None
            losses = Oracle(params).loss(x, c, pr, tfr, tfr, tfr, 0.1, [1, 0.5], eps[0], eps[1], lambda: 0.5)
            losses[0].backward()
            with torch.no_grad():
                clip_and_adam_step(plist, [p.grad for p in plist], m, v, step + 1, 1e-3)
            times.append(time.perf_counter() - t0)
        times = times[1:]
        out['tfr=%g' % tfr] = {'s_per_step_min': round(min(times), 3), 's_per_step_median': round(statistics.median(times), 3),
                               'samples_per_s_best': round(batch / min(times), 2)}
    best = out['tfr=1']
    ref_itself = None
    try:                                                    # the reference's own train.py step, timed in the build container (8 vCPU): read beside the port
        ref_itself = json.load(open(os.path.join(ROOT, 'profiles', 'r02_reference_cpu_timing.json')))
    except Exception:
        pass
    return {'value': best['samples_per_s_best'], 'unit': 'samples/s', 'cores': cores, 'kind': 'port',
            'reference_itself_build_container': ref_itself,
            'sample': '1 warm-up + %d timed train steps (fwd+bwd+clip+Adam) each at tfr=1 and tfr=0, batch %d, fp32, torch CPU '
                      'oracle/ptvae_oracle.py; value = best teacher-forced step' % (timed, batch), 'cases': out}


# ---- parity of the benched dtype against the reference-generated golden vectors (data only: expected outputs recorded from the
# reference by tests/golden/make_golden.py).  Outside the timed region.
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _full_params():
    import numpy as np
    from polyphonic_chord_texture_disentanglement_amd.synthetic import fill_state_dict
    with np.load(os.path.join(GOLDEN, 'full_shapes.npz'), allow_pickle=False) as f:
        shapes = {str(n): tuple(int(t) for t in s.strip('()').split(',') if t.strip()) for n, s in zip(f['names'], f['shapes'])}
    return fill_state_dict(shapes, 1234)


def golden_parity(precision, dev, case='full_tf1_b16', detail=False, via_loss=False):
    import numpy as np
    """{max_abs_dloss, rel_gradnorm_err, worst_tensor_gradnorm_rel_err}: the full init_model() geometry, teacher-forced step of
    `case` in `precision` against the losses / per-tensor gradient norms the reference produced on the same inputs"""
    from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE
    from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch
    with np.load(os.path.join(GOLDEN, case + '.npz'), allow_pickle=False) as f:
        g = {k: f[k] for k in f.files}
    x, c, pr = (torch.from_numpy(a).to(dev) for a in synth_batch(int(g['B']), int(g['data_seed'])))
    m = DisentangleVAE.init_model(dev)
    m.load_state_dict(_full_params())
    m.to(dev).set_precision(precision)
    m.eps_source = lambda name, shape, device: torch.from_numpy(g['eps_' + name]).to(device)
    m.zero_grad()
    if via_loss:        # the entry point the benchmark times: loss() = run + loss_function in one call (the decoder stops at the batch's last target)
        losses = m.loss(x, c, pr, 1., 1., 1., float(g['beta']), [float(w) for w in g['weights']])
    else:
        outs = m.run(x, c, pr, 1., 1., 1.)
        losses = m.loss_function(x, c, *outs, float(g['beta']), [float(w) for w in g['weights']])
    got = np.array([l.item() for l in losses])
    losses[0].backward()
    tot2, ref2, worst, worst_name = 0.0, 0.0, 0.0, None
    for k, p in m.named_parameters():
        gn, ref = float(p.grad.double().pow(2).sum().sqrt()), float(g['gnorm.' + k])
        tot2 += gn * gn
        ref2 += ref * ref
        e = abs(gn - ref) / max(ref, 1e-30)
        if e > worst:
            worst, worst_name = e, k
    res = {'fixture': 'tests/golden/%s.npz' % case, 'dtype': 'bf16' if precision == 'bf16' else 'f32',
           'max_abs_dloss': float(np.abs(got - g['losses']).max()), 'loss': float(got[0]), 'loss_ref': float(g['losses'][0]),
           'rel_gradnorm_err': abs(tot2 ** 0.5 - ref2 ** 0.5) / ref2 ** 0.5, 'worst_tensor_gradnorm_rel_err': worst}
    if detail:
        res['worst_tensor'] = worst_name
        res['dloss'] = [float(v) for v in (got - g['losses'])]
    return res


def _measure(step_fn, steps, warmup, repeats=2):
    """side figures (extras) only: seconds per step over `steps` steps after `warmup`, best of `repeats` timed blocks -- a block of four
    22-ms steps is halved by ONE caching-allocator growth (hipMalloc synchronises) right after the previous workload's cache was
    dropped; seen once as 9.2k instead of 22.8k samples/s.  The headline measurement (main()) times exactly K steps once, as the contract says."""
    for i in range(warmup):
        step_fn(i)
    best = None
    for r in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step_fn(warmup + r * steps + i)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / steps
        best = t if best is None else min(best, t)
    return best


def extras(dev, B, rank):
    """driver-visible side figures (never `value`): the other BASELINE configs and the trainer surface, a few steps each"""
    import random
    from polyphonic_chord_texture_disentanglement_amd.amc_dl import torch_plus as tp
    from polyphonic_chord_texture_disentanglement_amd.amc_dl.torch_plus.train_utils import kl_anealing
    from polyphonic_chord_texture_disentanglement_amd.dataset_loaders import DeviceBatcher, TrainingVAE
    from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE, LOSS_NAMES
    from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam
    from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch, synth_raw_bank
    out = {}

    def train_setup(prec):
        torch.manual_seed(0)
        m = DisentangleVAE.init_model(dev).to(dev).set_precision(prec)
        m.use_philox(7, 0)
        opt = FusedClipAdam(m.parameters(), lr=1e-3)
        data = [tuple(torch.from_numpy(a).to(dev) for a in synth_batch(B, 99 + i)) for i in range(2)]
        return m, opt, data

    def train_fn(m, opt, data, tfr):
        def fn(i):
            x, c, pr = data[i % 2]
            opt.zero_grad()
            o = m('train', x, c, pr, tfr1=tfr, tfr2=tfr, tfr3=tfr, beta=0.1, weights=[1, 0.5])
            o[0].backward()
            opt.clip_and_step(1.0)
        return fn

    import ctypes
    from polyphonic_chord_texture_disentanglement_amd._lib import lib as _lib_
    L = _lib_()

    def note_loop_record(rows):
        """the free-running note loop (ptv_free_note_loop: 15 note steps of every 16-sample panel per launch, 32 launches per forward) on the
        MFMA roofline by its algorithmic FLOPs -- and as what it is: a latency chain, microseconds per dependent note step"""
        cnt, ms, fl = ctypes.c_long(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
        L.ptv_prof_read_tag(7, ctypes.byref(cnt), ctypes.byref(ms), ctypes.byref(fl))
        if not cnt.value:
            return None
        tfs = fl.value / (ms.value * 1e-3) / 1e12
        return {'bound': 'mfma', 'kernel': 'note_loop_kernel / note_loop2_kernel (csrc/freerun.hip), B = %d rows' % rows, 'achieved': round(tfs, 2),
                'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': round(tfs / 2500.0, 5), 'launches': cnt.value, 'avg_us': round(ms.value / cnt.value * 1e3, 1),
                'us_per_note_step': round(ms.value / cnt.value * 1e3 / 15, 2),
                'bound_measured': 'latency: 480 dependent note steps per forward (gate products -> state all-gather -> pitch head -> argmax -> '
                                  'dur_hid -> 5 duration steps -> token embedding), weights streamed from L2 per panel'}

    def timed_with_note_loop(fn, steps, warm, rows):
        t_ = _measure(fn, steps, warm)
        L.ptv_prof_reset(); L.ptv_prof_enable(64)            # tag 7, two more steps outside the timing
        fn(0); fn(1)
        torch.cuda.synchronize()
        L.ptv_prof_enable(0)
        return t_, note_loop_record(rows)

    random.seed(7)
    m, opt, data = train_setup('bf16')
    t, nl = timed_with_note_loop(train_fn(m, opt, data, 0.0), 4, 2, B)
    out['train_free_running_tfr0'] = {'samples_per_s': round(B / t, 1), 'ms_per_step': round(t * 1e3, 2), 'batch': B, 'dtype': 'bf16',
                                      'note': "the reference's train.py schedule from its third batch on (SURVEY 0.4)", 'roofline_note_loop': nl}
    del m, opt, data
    torch.cuda.empty_cache()
    # BASELINE configs[4] names 1024 samples per GPU for this schedule: the step loop is a latency chain per 16-sample panel, so the larger
    # batch fills the chip (64 panels x 4 cluster members = one workgroup per CU)
    Bsave = B
    B = 1024
    try:
        random.seed(7)
        m, opt, data = train_setup('bf16')
        t, nl = timed_with_note_loop(train_fn(m, opt, data, 0.0), 4, 2, B)
        out['train_free_running_tfr0_b1024'] = {'samples_per_s': round(B / t, 1), 'ms_per_step': round(t * 1e3, 2), 'batch': B, 'dtype': 'bf16',
                                                'note': "configs[4]'s per-GPU batch on train.py's schedule from its third batch on (tfr = 0)",
                                                'roofline_note_loop': nl}
        del m, opt, data
    finally:
        B = Bsave
    torch.cuda.empty_cache()
    m, opt, data = train_setup('bf16')
    # the trainer surface: TrainingVAE.train() with the device-resident data path (raw piano-roll bank -> ptv_batch_transform),
    # train.py's schedulers, fused clip+Adam, one non-blocking 11-scalar log per batch
    pr_bank, ch_bank = synth_raw_bank(1024, 5)              # 12288 augmented samples: 24 batches per epoch (an epoch ends with a log flush = a sync)
    loader = DeviceBatcher(pr_bank, ch_bank, B, seed=1, device=dev, drop_last=True)
    nb = len(loader)

    class _L:
        train_loader, val_loader = loader, []
    pm = tp.LogPathManager(None, log_path_name=os.path.join(os.environ.get('TMPDIR', '/tmp'), 'ptvae_bench'))
    osch = tp.OptimizerScheduler(opt, tp.MinExponentialLR(opt, gamma=0.9999, minimum=1e-5), 1)
    ps = tp.ParameterScheduler(tfr1=tp.ConstantScheduler(1.), tfr2=tp.ConstantScheduler(1.), tfr3=tp.ConstantScheduler(1.),
                               beta=tp.TeacherForcingScheduler(0.1, 0., f=kl_anealing), weights=tp.ConstantScheduler([1, 0.5]))
    sw = tp.SummaryWriters(LOSS_NAMES, {'loss': None}, pm.writer_path)
    tr = TrainingVAE(dev, m, False, pm, _L, sw, osch, ps, 1)
    tr.train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / nb
    out['trainer_surface_teacher_forced'] = {'samples_per_s': round(B / t, 1), 'ms_per_step': round(t * 1e3, 2), 'batch': B,
                                             'note': 'TrainingVAE.train(): device batch transform + step + async logging, %d batches' % nb}
    del m, opt, tr
    torch.cuda.empty_cache()
    m, opt, data = train_setup('fp32')
    t = _measure(train_fn(m, opt, data, 1.0), 3, 1)
    out['train_teacher_forced_fp32'] = {'samples_per_s': round(B / t, 1), 'ms_per_step': round(t * 1e3, 2), 'batch': B, 'dtype': 'f32',
                                        'note': 'the <=1e-4 parity path (exact fp32 MFMA)'}
    del m, opt
    torch.cuda.empty_cache()
    # smaller per-GPU batches, where the host (not the GPU) bounds the eager step: eagerly enqueued vs replayed from ONE captured
    # hipGraph per step (graph_step.GraphedTrainStep: what TrainingInterface.train() uses by default for batch <= GRAPH_AUTO_MAX_BATCH = 64)
    from polyphonic_chord_texture_disentanglement_amd.graph_step import GraphedTrainStep
    for Bs in (64, 128, 256):
        torch.manual_seed(0)
        m = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16')
        m.use_philox(7, 0)
        opt = FusedClipAdam(m.parameters(), lr=1e-3)
        data = [tuple(torch.from_numpy(a).to(dev) for a in synth_batch(Bs, 99 + i)) for i in range(2)]
        with torch.autograd.set_multithreading_enabled(False):          # (as TrainingInterface.train() runs its backward passes)
            te = _measure(train_fn(m, opt, data, 1.0), 8, 3)
        gs = GraphedTrainStep(m, opt, Bs)
        gs(*data[0])
        t0 = time.perf_counter()
        tg = _measure(lambda i: gs(*data[i % 2]), 12, 2)
        out['train_teacher_forced_b%d' % Bs] = {
            'eager': {'samples_per_s': round(Bs / te, 1), 'ms_per_step': round(te * 1e3, 2)},
            'graph_replayed': {'samples_per_s': round(Bs / tg, 1), 'ms_per_step': round(tg * 1e3, 2)}, 'batch': Bs, 'dtype': 'bf16',
            'note': 'whole step (zero_grad, forward, backward, clip+Adam) as one captured hipGraph per step vs ~290 eager launches (backward pass on the calling thread, as the trainer runs it)'}
        del m, opt, gs, data
        torch.cuda.empty_cache()
    # the headline with every zero-skip switched off: the backward computes the (exactly zero) gradients of the padded note slots and
    # the note-summary GRU runs all 16 note positions of every row, as dense autograd would
    from polyphonic_chord_texture_disentanglement_amd import functional as F_
    F_.ZERO_SKIP = False
    try:
        m, opt, data = train_setup('bf16')
        t = _measure(train_fn(m, opt, data, 1.0), 6, 3)
        out['train_teacher_forced_dense_backward'] = {
            'samples_per_s': round(B / t, 1), 'ms_per_step': round(t * 1e3, 2), 'batch': B, 'dtype': 'bf16',
            'note': 'PTV_ZERO_SKIP=0: same results, EVERYTHING dense -- the headline (i) passes over note steps / tiles whose gradient is '
                    'exactly zero (decided on the gradients themselves) and over packed-sequence padding in the backward, and (ii) in '
                    'loss() stops the decoder forward at the batch\'s last note step that holds a target (the loss ignores the rest)'}
        del m, opt
    finally:
        F_.ZERO_SKIP = True
        F_.zero_skip_sync()
    torch.cuda.empty_cache()
    # ... and with only the forward dense (every note step computed, as run() on its own does), the backward zero-skipping
    F_.DEAD_STEPS = False
    try:
        m, opt, data = train_setup('bf16')
        t = _measure(train_fn(m, opt, data, 1.0), 6, 3)
        out['train_teacher_forced_full_forward'] = {
            'samples_per_s': round(B / t, 1), 'ms_per_step': round(t * 1e3, 2), 'batch': B, 'dtype': 'bf16',
            'note': 'PTV_DEAD_STEPS=0: the decoder forward computes all 15 note steps although loss() uses only those up to the last target'}
        del m, opt
    finally:
        F_.DEAD_STEPS = True
    torch.cuda.empty_cache()
    torch.manual_seed(0)
    m = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16')
    m.decoder.use_graph = True
    Bd = 2048
    z = torch.randn(Bd, 512, device=dev)

    def dec(i):
        with torch.no_grad():
            m.decoder(z, True, None, None, 0., 0.)
    t = _measure(dec, 3, 2)
    m.decoder.use_graph = False                              # (the note loop's events are recorded on eager launches)
    L.ptv_prof_reset(); L.ptv_prof_enable(64)
    dec(0)
    torch.cuda.synchronize()
    L.ptv_prof_enable(0)
    out['decode_free_running_b2048_graph'] = {'samples_per_s': round(Bd / t, 1), 'ms_per_decode': round(t * 1e3, 2), 'batch': Bd,
                                              'note': 'configs[3]: inference_decode step loop replayed from a hipGraph',
                                              'roofline_note_loop': note_loop_record(Bd)}
    return out


PMC_FILE = 'profiles/r06_pmc_pick.json'


def _roofline(lib, B, model, args, ms_per_step=None):
    """roofline of the step's dominant kernel FAMILY, measured live: HIP events on each launch's own stream (ptv_prof_*,
    include/ptvae_hip_debug.h) over the timed region.  Candidates: the weight-gradient family (ptv_wgrad / ptv_wgrad_batch: product +
    reduction launches; MFMA roofline), the notes GRU of the teacher-forced decoder as one row-partitioned launch per direction of time
    (csrc/notes_roles.hip; HBM roofline), the BPTT of the persistent small-M recurrences (MFMA).  The top-level record is the candidate
    with the most summed launch time per step (round-5 review: it used to be the notes forward whatever its share); the rest follow in
    `also`, then the whole step on nominal and on EXECUTED FLOPs.

    Weight-gradient family: algorithmic FLOPs of a launch = sum over its products of 2*M*N*K_live, K_live = the rows below the
    product's device-side limit (ptv_wgrad's k_top: note steps / note positions at which no gradient arrives are exact zeros and are
    passed over); the library reports the limited part separately and the live fraction of the benchmark batch is applied here
    (`k_live`).  `achieved` = those FLOPs / the summed launch durations.

    Notes GRU, algorithmic bytes per launch (DESIGN.md section 4, R = 32*B rows, H = 512, E = 128, T = 15):
      fwd  per step: GC bf16 R*3H*2 + fed token fp32 R*E*4 read; state bf16 R*H*2 + four gate planes bf16 4*R*H*2 written; once: W_hh,
           W_ih[:, Ht:] bf16, b_hh, the initial state.  min_bytes = the same without the gate planes.
      bwd  per step: gates 4*R*H*2 + arriving gradient R*H*2 + previous state bf16 R*H*2 read; dgi bf16 R*3H*2 + the n third of dgh
           R*H*2 written; once: W_hh^T bf16, dh0 fp32."""
    import ctypes
    R, H, E, T = 32 * B, model.decoder.dec_notes_hid_size, 128, 15
    from polyphonic_chord_texture_disentanglement_amd import functional as F_
    # loss() stops the forward launch at the batch's last note step that holds a target: bytes and FLOPs of the launch scale with it
    last = F_._LIVE.get('last_counts')
    skip = F_.DEAD_STEPS and F_.ZERO_SKIP
    Tl = min(T, int(last[2].item()) + 1) if (last is not None and skip) else T
    once_f = 3 * H * (H + E) * 2 + 3 * H * 4 + R * H * 4
    fwd_bytes = Tl * (R * 3 * H * 2 + R * E * 4 + R * H * 2 + 4 * R * H * 2) + once_f
    fwd_min = Tl * (R * 3 * H * 2 + R * E * 4 + R * H * 2) + once_f
    bwd_bytes = T * (4 * R * H * 2 + R * H * 2 + R * H * 2 + R * 3 * H * 2 + R * H * 2) + 3 * H * H * 2 + R * H * 4
    pmc = {}
    # HBM traffic / executed MFMA instructions from the PMC counters (separate rocprofv3 --pmc passes over this same command,
    # scripts/gpu_pmc.sh, corrected as MI355X_MICROARCH.md prescribes): read from the committed file, stamped with its commit
    pmc_path = os.path.join(ROOT, PMC_FILE)
    if B == 512 and args.precision == 'bf16' and os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path))
    src = ('%s @ %s' % (PMC_FILE, pmc.get('_commit'))) if pmc else None
    steps = max(1, args.steps)

    def read(tag):
        cnt, ms, fl = ctypes.c_long(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
        lib.ptv_prof_read_tag(tag, ctypes.byref(cnt), ctypes.byref(ms), ctypes.byref(fl))
        return cnt.value, ms.value, fl.value

    out = []
    # ---- weight-gradient family (MFMA roofline)
    cnt, ms, fl = read(5)
    if cnt:
        l15, l16 = ctypes.c_double(0.0), ctypes.c_double(0.0)
        lib.ptv_prof_read_limited(5, ctypes.byref(l15), ctypes.byref(l16))
        # live fractions of this batch: decoder products stop at the last note step with a target (Tl of 15); the note-summary GRU's
        # products at the longest note list of the batch (lengths = targets per (sample, step) + 1 for the <sos> slot, capped at 16)
        f15 = Tl / float(T) if F_.ZERO_SKIP else 1.0
        f16 = min(16, Tl + 1) / 16.0 if F_.ZERO_SKIP else 1.0
        # ... and, of those, the products over (note step, length-sorted row) also skip the dead 128-row blocks of every live step
        # (ptv_wgrad_job.seg_n): live fraction = sum of the steps' live prefixes / (15 R), from the same device array the products read
        lseg = ctypes.c_double(0.0)
        lib.ptv_prof_read_segmented(5, ctypes.byref(lseg))
        fseg, seg_t = f15, (getattr(F_, '_LAST_SEG_N', None) if F_.ZERO_SKIP else None)
        if lseg.value > 0 and seg_t is not None:
            fseg = float(seg_t.sum().item()) / (T * R)
        fl_live = fl - (l15.value - lseg.value) * (1.0 - f15) - lseg.value * (1.0 - fseg) - l16.value * (1.0 - f16)
        tfs, tfs_nom = fl_live / (ms * 1e-3) / 1e12, fl / (ms * 1e-3) / 1e12
        k = pmc.get('wgrad_family', {})
        out.append({'bound': 'mfma', 'kernel': 'weight-gradient family: wgrad_kernel / wgrad_batch_kernel + wgrad_reduce(_batch)_kernel (ptv_wgrad, '
                                               'ptv_wgrad_batch; every grad_W = grad_out^T . input of the step, bias sums fused)',
                    'achieved': round(tfs, 1), 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': round(tfs / 2500.0, 4),
                    'traffic': k.get('hbm_bytes_per_step'), 'traffic_unit': 'HBM bytes per step, all launches of the family (PMC)',
                    'traffic_source': src if k else None,
                    'algorithmic_flops_per_step': round(fl_live / steps), 'nominal_flops_per_step': round(fl / steps),
                    'k_live': {'decoder_note_steps': '%d of %d' % (Tl, T), 'frac15': round(f15, 4), 'frac16': round(f16, 4),
                               'frac_segmented': round(fseg, 4), 'segmented_share_of_nominal': round(lseg.value / fl, 4) if fl else None},
                    'achieved_on_nominal_flops': round(tfs_nom, 1),
                    'launches_per_step': round(cnt / steps, 1), 'avg_us': round(ms / cnt * 1e3, 1), 'total_ms': round(ms, 2),
                    'ms_per_step': round(ms / steps, 3),
                    'executed_mfma_tflop_per_step_pmc': k.get('executed_tflop_per_step'), 'mfma_busy_frac_pmc': k.get('mfma_busy_frac'),
                    'operand_bytes_note': 'products with M or N <= 135 are HBM-bound (3-4.4 TB/s standalone, profiles/r05_bench_wgrad.txt); the '
                                          'family figure is FLOP-weighted and dominated by the deep products',
                    'note': 'event-timed per call (a batch = one product launch + one reduction launch) on its own stream inside the timed region, '
                            'in situ: the calls overlap each other and the latency chains, so the summed time exceeds their share of the step'})
    # ---- notes GRU (HBM roofline)
    kernels = [(3, 'notes_fwd_kernel', fwd_bytes, fwd_min)]
    if not F_.ZERO_SKIP:                                    # with the zero-skip on, the BPTT launch moves a data-dependent share of
        kernels.insert(0, (4, 'notes_bwd_kernel', bwd_bytes, bwd_bytes))   # these bytes (late note steps without gradient are passed over)
    for tag, name, nbytes, nmin in kernels:
        cnt, ms, fl = read(tag)
        if cnt == 0:
            continue
        steps_run = Tl if tag == 3 else T
        fl *= steps_run / T                                  # (the library counts the launch's FLOPs for all T steps)
        avg_s = ms / cnt * 1e-3
        gbs, tfs = nbytes / avg_s / 1e9, fl / cnt / avg_s / 1e12
        k = pmc.get(name, {})
        out.append({'bound': 'hbm', 'kernel': '%s (dec_notes_gru, R=%d rows x %d of T=%d steps in one launch%s)' % (
                        name, R, steps_run, T, ': loss() stops at the last note step with a target' if steps_run < T else ''),
                    'achieved': round(gbs, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(gbs / 8000.0, 4),
                    'traffic': k.get('hbm_bytes_per_launch'), 'traffic_source': src if k else None,
                    'algorithmic_bytes': nbytes, 'min_bytes': nmin, 'frac_on_min_bytes': round(nmin / avg_s / 1e9 / 8000.0, 4),
                    'launches': cnt, 'avg_us': round(avg_s * 1e6, 1), 'total_ms': round(ms, 2), 'ms_per_step': round(ms / steps, 3),
                    'mfma': {'achieved': round(tfs, 1), 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': round(tfs / 2500.0, 4),
                             'busy_frac_pmc': k.get('mfma_busy_frac')},
                    'flop_per_byte': round(fl / cnt / nbytes, 1), 'balance_flop_per_byte': 312.5,
                    'bound_measured': 'per-CU vector-memory path (in-order across waves), neither HBM pins nor MFMA',
                    'bound_evidence': 'profiles/r05_notes_roles_ablation.txt (no weights / no activation streams / neither), '
                                      'profiles/r05_tcp_order.txt (one HBM-missing wave slows the CU\'s L2-hit stream 1.46x, four 3.6x)',
                    'note': 'in situ: the launch shares the GPU with the weight-gradient products on sibling streams'})
    # ---- BPTT of the persistent small-M recurrences (MFMA roofline by FLOPs; bound by its per-step hand-offs)
    cnt, ms, fl = read(6)
    if cnt:
        tfs = fl / (ms * 1e-3) / 1e12
        out.append({'bound': 'mfma', 'kernel': 'pgru_bwd_sk_kernel / pgru_bwd_kernel (BPTT of the small-M persistent recurrences)',
                    'achieved': round(tfs, 1), 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': round(tfs / 2500.0, 4), 'traffic': None,
                    'launches': cnt, 'total_ms': round(ms, 2), 'ms_per_step': round(ms / steps, 3), 'flops': '2*chains*M*3H*H*T',
                    'bound_measured': 'hand-off latency (two L2 exchanges per recurrent step), DESIGN.md section 4'})
    if not out:
        return None
    out.sort(key=lambda r: -r['total_ms'])
    roof = out[0]
    roof['selected_by'] = 'largest summed launch time per step among the event-timed families'
    roof['also'] = out[1:]
    if ms_per_step:
        tfs = GFLOP_PER_SAMPLE_TRAIN * 1e9 * B / (ms_per_step * 1e-3) / 1e12
        roof['also'].append({'bound': 'mfma', 'kernel': 'whole train step, NOMINAL work (6.15 GFLOP per sample, SURVEY.md 8d: the exactly-zero work '
                                                        'the step passes over is counted)', 'achieved': round(tfs, 1),
                             'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': round(tfs / 2500.0, 4)})
        ex = pmc.get('_step', {}).get('executed_tflop_per_step')
        if ex:
            tfs = ex * 1e12 / (ms_per_step * 1e-3) / 1e12
            roof['also'].append({'bound': 'mfma', 'kernel': 'whole train step, EXECUTED MFMA work (sum of SQ_INSTS_MFMA x 16384 FLOP over the step\'s '
                                                            'kernels, PMC)', 'achieved': round(tfs, 1), 'peak': 2500.0,
                                 'unit': 'TFLOP/s', 'frac': round(tfs / 2500.0, 4), 'executed_tflop_per_step': ex, 'traffic_source': src})
    return roof


def _launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh children of this script, one per GPU (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, as `torch.distributed.run` would), BEFORE this process makes any GPU call -- it
    never does: the parent only waits.  Rank 0 inherits stdout (its ONE JSON line is this command's output), the other ranks'
    stdout goes to stderr.  Returns the exit code: non-zero when any rank failed (the others are then stopped).
    Replaces nn.DataParallel's in-process replication (reference amc_dl/torch_plus/module.py:67-68)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ.get('MASTER_PORT', str(port)), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc, live = 0, list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in live:                                   # one rank died: the others would wait for it in a collective forever
                    q.terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=512, help='per-GPU batch (weak scaling)')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true', help='skip the golden-vector parity block of the benched dtype')
    ap.add_argument('--no-extras', action='store_true', help='skip the side figures (tfr=0 train, trainer surface, fp32, decode)')
    ap.add_argument('--graph', action='store_true', help='decode mode: replay the step loop from a captured hipGraph')
    ap.add_argument('--tfr', type=float, default=1.0, help='teacher-forcing ratio (1 = configs[1]; 0 = free-running training)')
    ap.add_argument('--mode', default='train', choices=['train', 'decode'],
                    help="'decode' = configs[3]: free-running inference_decode samples/s")
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(_launch_ranks(args.gpus))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        ndev = torch.cuda.device_count()
        backend = os.environ.get('PTV_DIST_BACKEND', 'nccl')     # 'gloo' lets a 1-GPU box exercise the N>1 flow
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
        local_rank = local_rank % max(ndev, 1)
    assert world == args.gpus, '--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or without a launcher' % (args.gpus, world, args.gpus)
    if os.environ.get('PTV_BENCH_LAUNCH_ONLY') == '1':
        # launch-path check without a GPU (tests/test_dist_gloo.py): rendezvous, barrier, MAX over ranks, rank 0's one JSON line
        import torch.distributed as dist
        t0 = time.perf_counter()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({'launch_only': True, 'n_gpus': world, 'max_over_ranks': float(t.item()), 'steps': args.steps}), flush=True)
        dist.destroy_process_group()
        return
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    from polyphonic_chord_texture_disentanglement_amd import _lib
    from polyphonic_chord_texture_disentanglement_amd.amc_dl.torch_plus import MinExponentialLR
    from polyphonic_chord_texture_disentanglement_amd.dist import GradSync
    from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE
    from polyphonic_chord_texture_disentanglement_amd.optim import FusedClipAdam
    from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch
    lib = _lib.lib()

    torch.manual_seed(0)                                   # identical weights on every rank
    model = DisentangleVAE.init_model(dev).to(dev).set_precision(args.precision)
    opt = FusedClipAdam(model.parameters(), lr=1e-3)
    from polyphonic_chord_texture_disentanglement_amd.optim import reserve_step_memory
    reserve_step_memory(args.batch, dev)                       # allocator warm-up (setup): no hipMalloc -- a device-wide sync -- inside the steps
    model.decoder.use_graph = args.graph
    sched = MinExponentialLR(opt, gamma=0.9999, minimum=1e-5)
    sync = GradSync(model, opt) if world > 1 else None
    if sync is not None:
        sync.timing = True

    B = args.batch
    nb = 2
    batches = []
    for i in range(nb):                                    # disjoint data per rank (SURVEY §8d seeds)
        x, c, pr = synth_batch(B, 1234 + rank * 10 ** 6 + i)
        batches.append(tuple(torch.from_numpy(a).to(dev) for a in (x, c, pr)))
    import random
    gen = torch.Generator(device=dev).manual_seed(7 + rank)     # decode mode only
    model.use_philox(seed=7, sample_offset=rank * B)           # eps keyed by the GLOBAL sample index: sharding-invariant
    random.seed(7)                                             # teacher-forcing coins: one stream shared by all ranks

    def step(i):
        x, c, pr = batches[i % nb]
        opt.zero_grad()
        if args.mode == 'decode':
            zc = torch.randn(B, 256, device=dev, generator=gen)
            zr = torch.randn(B, 256, device=dev, generator=gen)
            with torch.no_grad():
                model.decoder(torch.cat([zc, zr], -1), True, None, None, 0., 0.)
            return (model.decoder.last_xhat.sum().float(),)
        out = model('train', x, c, pr, tfr1=args.tfr, tfr2=args.tfr, tfr3=args.tfr, beta=0.1, weights=[1, 0.5])
        out[0].backward()
        if sync is not None:
            sync.all_reduce_grads()
        opt.clip_and_step(1.0)
        sched.step()
        return out

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        out = step(i)
        if rank == 0:
            torch.cuda.synchronize()
            print('[bench] warmup step %d done' % i, file=sys.stderr, flush=True)
    import gc
    from polyphonic_chord_texture_disentanglement_amd.optim import freeze_gc, unfreeze_gc
    freeze_gc()                                            # a full collection (~100 ms: it walks the whole module tree) must not fall into
                                                           # the timed steps; what is alive now moves to the permanent generation
    barrier()
    lib.ptv_prof_reset()
    lib.ptv_prof_config(32 * B, model.decoder.dec_notes_hid_size)
    # tags 3, 4: the row-partitioned notes GRU, forward and BPTT; 5: the weight-gradient family (~20 calls per step since the products are
    # batched); 6: BPTT of the persistent recurrences -- all event-timed on their own streams INSIDE the timed region
    lib.ptv_prof_enable(4 | 8 | 16 | 32)
    opt.throttle_wait_s = 0.0
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    t_host = time.perf_counter() - t0                      # all K steps enqueued (host side of the pipeline) ...
    t_wait = getattr(opt, 'throttle_wait_s', 0.0)          # ... of which the optimiser spent this waiting for the GPU (two steps in flight at most)
    barrier()
    dt = time.perf_counter() - t0
    lib.ptv_prof_enable(0)
    rep = sync.exchange_report() if world > 1 else None        # (the exchange figures of the TIMED steps: taken before the extra ones below)
    dp = None
    if world > 1:
        import torch.distributed as dist
        # what a first N-GPU run needs to see where the time went: every rank's own step time, the part of the gradient exchange the
        # backward pass did not hide (event-timed on the step's stream), what left early and what was left for the end
        cpu_backend = dist.get_backend() != 'nccl'
        mine = torch.tensor([dt / args.steps * 1e3, rep['exposed_allreduce_ms'] or 0.0], dtype=torch.float64,
                            device='cpu' if cpu_backend else dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        dp = dict(rep, per_rank_ms_per_step=[round(float(a[0]), 3) for a in allr],
                  per_rank_exposed_allreduce_ms=[round(float(a[1]), 3) for a in allr],
                  note='exposed_allreduce_ms = GPU time of all_reduce_grads() on the step stream per step (waits for the slices that '
                       'left during the backward pass + the all-reduce of the remainder); a ring all-reduce of the whole bucket over '
                       'xGMI is ~1.25 ms (SURVEY 8e)')
        dp['exposed_allreduce_ms'] = None if rep['exposed_allreduce_ms'] is None else round(rep['exposed_allreduce_ms'], 3)
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(out[0].item())

    if rank == 0:
        import ctypes
        ms_per_step = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        roof = _roofline(lib, B, model, args, ms_per_step) if args.mode == 'train' and args.tfr >= 1.0 else None
        if args.mode == 'decode':
            workload = 'configs[3]: free-running PtvaeDecoder sampling (inference_decode), batch=%d, 32x15x(1+5) step loop' % B
        elif args.tfr >= 1.0:
            workload = ('configs[%d]: %dxMI355X batch=%d/GPU%s, %s MFMA GRU/Linear, z_dim=256+256, teacher-forced decoder '
                        '(tfr=1), fwd+bwd+clip+Adam' % (1 if world == 1 else 2, world, B,
                                                        '' if world == 1 else ' (DDP, RCCL grad all-reduce)', args.precision))
        else:
            workload = 'train step with teacher-forcing ratio %.2f (step-loop decoder), batch=%d/GPU, fwd+bwd+clip+Adam' % (args.tfr, B)
        res = {'metric': '2-bar piano-roll samples/sec (train step)' if args.mode == 'train' else '2-bar piano-roll samples/sec (free-running decode)', 'value': round(value, 1), 'unit': 'samples/s',
               'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': 'bf16' if args.precision == 'bf16' else 'f32', 'data': 'synthetic',
               'config': {'workload': workload,
                          'per_gpu_batch': B, 'global_batch': B * world, 'parallelism': 'dp%d' % world},
               'step_tflops': round(value * GFLOP_PER_SAMPLE_TRAIN / 1e3, 2),
               'step_tflops_note': 'NOMINAL: 6.15 GFLOP per sample x samples/s (work the step passes over as exactly zero is counted); the executed '
                                   'rate is roofline.also[whole train step, EXECUTED MFMA work]',
               'final_loss': round(loss, 4),
               'host_enqueue_ms_per_step': round((t_host - t_wait) / args.steps * 1e3, 3),
               'host_wait_ms_per_step': round(t_wait / args.steps * 1e3, 3), 'roofline': roof}
        if args.mode == 'train' and args.tfr >= 1.0:
            from polyphonic_chord_texture_disentanglement_amd import functional as F_
            res['exact_work_elision'] = {
                'zero_skip_backward': bool(F_.ZERO_SKIP), 'dead_note_steps_forward': bool(F_.DEAD_STEPS and F_.ZERO_SKIP),
                'note': 'same losses / gradients / parameter updates as the dense step: the backward passes over work whose result is '
                        'exactly zero, and loss() does not compute decoder outputs that the loss ignores (padded note slots after the batch\'s '
                        'last target); extra.train_teacher_forced_full_forward = forward dense, extra.train_teacher_forced_dense_backward '
                        '= everything dense (DESIGN.md section 4)'}
        if dp is not None:
            res['data_parallel'] = dp
        from polyphonic_chord_texture_disentanglement_amd.functional import persist_check
        persist_check()                                        # a persistent launch that gave up would have invalidated the run
        if world == 1 and args.mode == 'train' and not args.no_extras:
            unfreeze_gc()                                      # (the headline model may be collected again: the side figures build their own)
            del model, opt
            gc.collect()
            torch.cuda.empty_cache()
            try:
                res['extra'] = extras(dev, B, rank)
            except Exception as e:                             # side figures must never cost the headline line
                res['extra'] = {'error': repr(e)}
        if world == 1 and args.mode == 'train' and not args.no_parity:
            try:
                res['parity'] = {'benched': golden_parity(args.precision, dev, via_loss=True), 'run_then_loss_function': golden_parity(args.precision, dev), 'fp32_path': golden_parity('fp32', dev) if args.precision != 'fp32' else None,
                                 'bar': 'north_star: losses within 1e-4 of the CPU reference -- held by the fp32 path (fp32_path / by_batch.*.fp32); the benched bf16 dtype '
                                        '(bf16 MFMA operands + bf16-stored saved tensors) is held to 3e-4 by the tests (tests/test_gpu_model_wide.py), observed 2e-5 .. 1e-4'}
                # the same at B = 4 and at the BENCHED batch (round 4: full_tf1_b512.npz, produced by the reference at B = 512)
                res['parity']['by_batch'] = {
                    'b%d' % b_: {p_: golden_parity(p_, dev, case='full_tf1_b%d' % b_) for p_ in dict.fromkeys((args.precision, 'fp32'))}
                    for b_ in (4, 512)}
            except Exception as e:
                res['parity'] = {'error': repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
