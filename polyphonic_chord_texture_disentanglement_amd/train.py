"""Training entry point with the reference `train.py`'s constants and wiring (train.py:16-72); the two
deltas SURVEY.md §0.2 calls for: the texture encoder is `TextureEncoder(256, 1024, 256)` (the wiring
that runs) and batches carry three tensors.  Data is the synthetic generator (no POP909 here).

    python -m polyphonic_chord_texture_disentanglement_amd.train [--epochs 1 --batch 128 --precision bf16]
    python -m torch.distributed.run --nproc-per-node N -m polyphonic_chord_texture_disentanglement_amd.train
"""
import argparse
import os
import random

import torch

from .amc_dl.torch_plus import (ConstantScheduler, LogPathManager, MinExponentialLR, OptimizerScheduler,
                                ParameterScheduler, SummaryWriters, TeacherForcingScheduler)
from .amc_dl.torch_plus.train_utils import kl_anealing
from .dataset_loaders import SEED, MusicDataLoaders, TrainingVAE
from .model import DisentangleVAE
from .optim import FusedClipAdam
from .ptvae import PtvaeDecoder, RnnDecoder, RnnEncoder, TextureEncoder

batch_size = 128
n_epoch = 6
clip = 1
weights = [1, 0.5]
beta = 0.1
tf_rates = [(0.6, 0), (0.5, 0), (0.5, 0)]
lr = 1e-3
name = 'disvae-nozoth'
writer_names = ['loss', 'recon_loss', 'pl', 'dl', 'kl_loss', 'kl_chd', 'kl_rhy', 'chord_loss', 'root_loss',
                'chroma_loss', 'bass_loss']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--epochs', type=int, default=n_epoch)
    ap.add_argument('--batch', type=int, default=batch_size)
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--teacher-forced', action='store_true',
                    help='hold tfr=1 (the published schedule decays to free-running after 2 steps)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        torch.distributed.init_process_group('nccl', device_id=device)

    rank = int(os.environ.get('RANK', 0))
    torch.manual_seed(0)                      # identical initial weights on every rank (GradSync also broadcasts rank 0's)
    random.seed(7)                            # teacher-forcing coins are per-step decisions for the WHOLE batch: one shared stream
    chd_encoder = RnnEncoder(36, 1024, 256)
    rhy_encoder = TextureEncoder(256, 1024, 256)
    chd_decoder = RnnDecoder(z_dim=256)
    pt_decoder = PtvaeDecoder(note_embedding=None, dec_dur_hid_size=64, z_size=512)
    model = DisentangleVAE(name, device, chd_encoder, rhy_encoder, pt_decoder, chd_decoder).to(device)
    model.set_precision(args.precision)
    model.use_philox(seed=7, sample_offset=rank * args.batch)      # eps keyed by the global sample index: sharding-invariant

    data_loaders = MusicDataLoaders.get_loaders(SEED + rank * 10 ** 7, bs_train=args.batch,
                                                bs_val=args.batch, portion=8, shift_low=-6, shift_high=5, num_bar=2,
                                                contain_chord=True)
    # one result directory per job: rank 0 names it (second-resolution timestamp) and writes into it; the other ranks get a
    # private scratch directory so that nothing they might emit collides with rank 0's checkpoints and scalars
    log_path_mng = LogPathManager(None) if rank == 0 else LogPathManager(None, log_path_name=os.path.join(
        os.environ.get('TMPDIR', '/tmp'), 'ptvae_rank%d' % rank))
    optimizer = FusedClipAdam(model.parameters(), lr=lr)
    scheduler = MinExponentialLR(optimizer, gamma=0.9999, minimum=1e-5)
    optimizer_scheduler = OptimizerScheduler(optimizer, scheduler, clip)
    summary_writers = SummaryWriters(writer_names, {'loss': None}, log_path_mng.writer_path)
    if args.teacher_forced:
        tfr = [ConstantScheduler(1.0) for _ in range(3)]
    else:
        tfr = [TeacherForcingScheduler(*r) for r in tf_rates]
    param_scheduler = ParameterScheduler(tfr1=tfr[0], tfr2=tfr[1], tfr3=tfr[2],
                                         beta=TeacherForcingScheduler(beta, 0., f=kl_anealing),
                                         weights=ConstantScheduler(weights))
    training = TrainingVAE(device, model, world > 1, log_path_mng, data_loaders, summary_writers, optimizer_scheduler,
                           param_scheduler, args.epochs)
    if training.grad_sync is not None:
        training.grad_sync.optimizer = optimizer
    training.run()


if __name__ == '__main__':
    main()
