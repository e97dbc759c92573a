"""Training entry points of the DoubleConv-family models on libadn -- the MI355X counterparts of
/root/reference/train_rgb_depth.py, train_binaural_attention.py and train_adabins_distillation.py.

Each ``main_*`` keeps the reference script's flag names and defaults, its directory conventions
(``./checkpoints/<experiment>/epoch_XXXX.pth`` + ``best_model.pth`` with the keys ``epoch, model_state_dict,
optimizer_state_dict, ...``; AdaBins: ``./results/<experiment>/``), its loop order (train epoch -> validation with
``compute_errors`` -> scheduler step -> checkpoints) and its loss / optimizer / scheduler choices; the per-batch body
(forward, loss, backward, [clip], optimizer step) is ONE fused libadn step (engine.FusedTrainer /
adabins_engine.AdaBinsTrainer).  Extra flags: ``--synthetic N`` (BatVision-shaped random items; no
dataset is needed), ``--precision bf16|f32|mxfp8`` (mxfp8: block-scaled fp8 3 x 3 conv forward / input-gradient GEMMs, BASELINE config 5).  wandb / visualisation plumbing is out of scope.
Thin launchers with the reference's script names live next to this file.

Multi-GPU: where the reference wraps the model in ``nn.DataParallel(gpu_ids)``, launch these with torchrun (one
process per GPU, ``--batch_size`` is per GPU): ddp.GradientAllReducer reproduces DataParallel's single global-batch
loss and SUM-reduces the gradients over RCCL; BatchNorm stays per replica; rank 0 prints and writes the checkpoints.
"""
import argparse
import math
import os
import time

import torch
from torch.utils.data import DataLoader, Dataset

from .config_loader import load_config
from .engine import FusedTrainer
from .utils_criterion import compute_errors


class SyntheticDepthItems(Dataset):
    """kind 'rgb': (image[3], depth); 'audio': (audio[2], depth); 'both': (audio[2], image[3], depth)."""

    def __init__(self, n, size, max_depth, kind):
        self.n, self.size, self.max_depth, self.kind = n, size, max_depth, kind

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(1234 + idx)
        S = self.size
        audio, image = torch.rand(2, S, S, generator=g), torch.rand(3, S, S, generator=g)
        gt = self.max_depth * torch.rand(1, S, S, generator=g)
        gt[gt < 0.1 * self.max_depth] = 0.0
        return {'rgb': (image, gt), 'audio': (audio, gt), 'both': (audio, image, gt)}[self.kind]


def lr_at(epoch, kind, base_lr, nb_epochs, eta_min=0.0):
    """Learning rate DURING epoch ``epoch`` (0-based) of torch's CosineAnnealingLR(T_max=nb_epochs, eta_min) /
    StepLR(50, 0.5) / none, stepped once per epoch as the reference does."""
    if kind == 'cosine':
        return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / nb_epochs)) / 2
    if kind == 'step':
        return base_lr * (0.5 ** (epoch // 50))
    return base_lr


def _common_flags(p, lr, batch):
    p.add_argument('--dataset', type=str, default='batvisionv2', choices=['batvisionv1', 'batvisionv2'])
    p.add_argument('--batch_size', type=int, default=batch)
    p.add_argument('--num_workers', type=int, default=4)
    p.add_argument('--base_channels', type=int, default=64)
    p.add_argument('--bilinear', action='store_true', default=True)
    p.add_argument('--learning_rate', type=float, default=lr)
    p.add_argument('--nb_epochs', type=int, default=200)
    p.add_argument('--optimizer', type=str, default='AdamW', choices=['Adam', 'AdamW', 'SGD'])
    p.add_argument('--weight_decay', type=float, default=0.01)
    p.add_argument('--scheduler', type=str, default='cosine', choices=['none', 'cosine', 'step'])
    p.add_argument('--checkpoints', type=int, default=None)
    p.add_argument('--save_frequency', type=int, default=2)
    p.add_argument('--use_wandb', action='store_true')
    p.add_argument('--wandb_project', type=str, default='batvision-depth-estimation')
    p.add_argument('--experiment_name', type=str, default=None)
    p.add_argument('--device', type=str, default='cuda')
    p.add_argument('--seed', type=int, default=42)
    p.add_argument('--synthetic', type=int, default=0, help='train on N BatVision-shaped random items')
    p.add_argument('--precision', type=str, default='bf16', choices=['bf16', 'f32', 'mxfp8'])


PRECISIONS = {'bf16': torch.bfloat16, 'f32': torch.float32, 'mxfp8': torch.float8_e4m3fn}


def _engine_of(model, args):
    """The model's engine in the precision of ``--precision``.  The precision has to be fixed BEFORE the trainer is built:
    ``model.engine()`` returns a new engine when ``compute_dtype`` changes, and a trainer built on the previous one would
    train in the old precision while validation runs the new engine -- two engines re-binding the parameters into their
    own flat buffers in turn, which resets the optimizer state at every epoch."""
    model.compute_dtype = PRECISIONS[args.precision]
    return model.engine()


def _dist():
    """(rank, world, local_rank, reducer): torch.distributed from the torchrun environment, RCCL reducer when world > 1."""
    from . import ddp as addp
    rank, world, local = addp.init_from_env()
    return rank, world, local, (addp.GradientAllReducer() if world > 1 else None)


def _device(args, local=None):
    if not torch.cuda.is_available():
        raise RuntimeError('training runs on libadn HIP kernels: no HIP device is visible (there is no CPU path)')
    dev = torch.device(args.device)
    if local is not None and torch.cuda.device_count() > 1:
        dev = torch.device('cuda', local)
    elif dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    torch.cuda.set_device(dev)
    return dev


def _loaders(cfg, args, kind, rank=0, world=1):
    """(train loader, val loader, sampler, batched device front-end or None).

    Real datasets are built with ``frontend='raw'``: DataLoader workers only read files (a forked worker must never
    touch the HIP device the parent has initialised); the STFT / mel / resize of the whole batch runs in the parent
    through GpuAudioFrontend, inside the step and forward lambdas (what train.py does too)."""
    fe = None
    if args.synthetic:
        S, md = cfg.dataset.images_size, cfg.dataset.max_depth
        train, val = SyntheticDepthItems(args.synthetic, S, md, kind), SyntheticDepthItems(max(1, args.synthetic // 4), S, md, kind)
        workers = 0
    else:
        if kind == 'both':
            raise NotImplementedError('the (audio, image, depth) dataset variant needs OpenCV for the camera frames; '
                                      'use --synthetic in this image')
        from .dataloader.utils_dataset import GpuAudioFrontend
        if cfg.dataset.name == 'batvisionv1':
            from .dataloader.BatvisionV1_Dataset import BatvisionV1Dataset as DS
            train = DS(cfg, cfg.dataset.annotation_file_train, frontend='raw')
            val = DS(cfg, cfg.dataset.annotation_file_val, frontend='raw')
            mode = 'bv1'
        else:
            from .dataloader.BatvisionV2_Dataset import BatvisionV2Dataset as DS
            img = kind == 'rgb'
            train = DS(cfg, cfg.dataset.annotation_file_train, use_image=img, frontend='raw')
            val = DS(cfg, cfg.dataset.annotation_file_val, use_image=img, frontend='raw')
            mode = GpuAudioFrontend.bv2_mode(cfg.dataset.audio_format, cfg.dataset.max_depth)
        if kind == 'audio' and 'waveform' not in cfg.dataset.audio_format:
            fe = GpuAudioFrontend(mode, cfg.dataset.images_size)
        elif kind == 'rgb':                 # raw decoded frames (uint8 BGR) -> [B,3,S,S] RGB in [0,1] on the device
            from .dataloader.utils_dataset import GpuImageTransform
            fe = GpuImageTransform(cfg.dataset.images_size)
        workers = args.num_workers
    sampler = torch.utils.data.distributed.DistributedSampler(train, world, rank, shuffle=True) if world > 1 else None
    tl = DataLoader(train, batch_size=args.batch_size, shuffle=sampler is None, sampler=sampler, num_workers=workers,
                    pin_memory=True, drop_last=True)
    vl = DataLoader(val, batch_size=args.batch_size, shuffle=False, num_workers=workers, pin_memory=True)
    return tl, vl, sampler, fe


def _to_device(batch, dev, fe):
    batch = [t.to(dev, non_blocking=True) for t in batch]
    if fe is not None:
        batch[0] = fe(batch[0])            # raw waveforms [B,2,T] -> network input [B,2,S,S], one libadn call per batch
    return batch


def _validate(model, loader, dev, forward, fe=None):
    model.eval()
    errs = {k: [] for k in ('abs_rel', 'rmse', 'delta1', 'delta2', 'delta3')}
    with torch.no_grad():
        for batch in loader:
            batch = _to_device(batch, dev, fe)
            pred = forward(model, batch)
            abs_rel, rmse, d1, d2, d3, _, _ = compute_errors(batch[-1], pred)
            for k, v in zip(errs, (abs_rel, rmse, d1, d2, d3)):
                errs[k].append(v)
    model.train()
    return {k: float(sum(v) / max(1, len(v))) for k, v in errs.items()}


def _run(args, cfg, model, trainer, kind, step, forward, exp, ckpt_root='checkpoints', ckpt_fmt='epoch_{:04d}.pth',
         on_epoch=None, dist_info=(0, 1, None, None)):
    rank, world, local, reducer = dist_info
    dev = _device(args, local if world > 1 else None)
    model.compute_dtype = PRECISIONS[args.precision]
    assert trainer.engine is model.engine(), 'build the trainer on _engine_of(model, args)'
    model = model.to(dev).train()
    tl, vl, sampler, fe = _loaders(cfg, args, kind, rank, world)
    ckpt_dir = os.path.join(ckpt_root, exp)
    os.makedirs(ckpt_dir, exist_ok=True)
    say = print if rank == 0 else (lambda *a, **k: None)
    start, ck = 0, None
    if args.checkpoints:
        path = os.path.join(ckpt_dir, ckpt_fmt.format(args.checkpoints))
        if os.path.exists(path):
            ck = torch.load(path, map_location=dev)
            model.load_state_dict(ck['model_state_dict'])
            start = ck['epoch']
            say(f'Loaded checkpoint from epoch {start}')
    # Order matters under the reducer: bind the flat buffers ONCE, replicate rank 0's weights (DataParallel.replicate),
    # and only then restore the optimizer state -- a second bind_parameters() would re-allocate flat_p / flat_g, make
    # the trainer re-run its setup (zeroing the restored Adam moments) and leave the reducer on the stale gradients.
    eng = model.engine()
    if not eng._bound():
        eng.bind_parameters()
    if reducer is not None:
        reducer.broadcast_parameters(eng.flat_p)
    if ck is not None and isinstance(ck.get('optimizer_state_dict'), dict):
        trainer.load_state_dict(ck['optimizer_state_dict'], dev)
    best = float('inf')
    eta_min = getattr(args, 'eta_min', 0.0)
    for epoch in range(start, args.nb_epochs):
        trainer.lr = lr_at(epoch, args.scheduler, args.learning_rate, args.nb_epochs, eta_min)
        if on_epoch is not None:
            on_epoch(epoch + 1)
        if sampler is not None:
            sampler.set_epoch(epoch)
        t0, losses = time.time(), []
        for i, batch in enumerate(tl):
            loss = step(trainer, _to_device(batch, dev, fe))
            losses.append(loss.detach().clone())
            if (i + 1) % 10 == 0:
                say(f'Epoch [{epoch + 1}/{args.nb_epochs}] Batch [{i + 1}/{len(tl)}] Loss: {losses[-1].item():.4f}')
        train_loss = torch.stack(losses).mean().item() if losses else float('nan')
        errs = _validate(model, vl, dev, forward, fe)
        say(f'Epoch [{epoch + 1}/{args.nb_epochs}] train loss {train_loss:.4f}  val RMSE {errs["rmse"]:.4f} '
              f'ABS_REL {errs["abs_rel"]:.4f} Delta1 {errs["delta1"]:.4f}  lr {trainer.lr:.2e}  {time.time() - t0:.1f}s')
        state = {'epoch': epoch + 1, 'model_state_dict': model.state_dict(), 'optimizer_state_dict': trainer.state_dict(),
                 'train_loss': train_loss, 'val_errors': errs}
        if (epoch + 1) % args.save_frequency == 0 and rank == 0:
            torch.save(state, os.path.join(ckpt_dir, ckpt_fmt.format(epoch + 1)))
        if errs['rmse'] < best:
            best = errs['rmse']
            if rank == 0:
                torch.save(dict(state, best_rmse=best), os.path.join(ckpt_dir, 'best_model.pth'))
    return model


# ---- train_rgb_depth.py ---------------------------------------------------------------------------------------------
def main_rgb(argv=None):
    """/root/reference/train_rgb_depth.py:91-463: RGBDepthNet, DepthLoss (L1 + 0.1 TV; the reference builds it with its
    defaults whatever --lambda_* say, :43-87, 252), AdamW(lr 1e-4, wd 0.01), cosine schedule, no clipping."""
    from .models.rgb_depth_model import create_rgb_depth_model
    p = argparse.ArgumentParser(description='Train RGB depth estimation model on Batvision dataset (MI355X)')
    _common_flags(p, 1e-4, 64)
    p.add_argument('--lambda_l1', type=float, default=1.0)
    p.add_argument('--lambda_smooth', type=float, default=0.1)
    args = p.parse_args(argv)
    cfg = load_config(dataset_name=args.dataset, model_name='unet_baseline', mode='train', experiment_name=args.experiment_name)
    exp = args.experiment_name or f'rgb_depth_{args.dataset}_BS{args.batch_size}_Lr{args.learning_rate}_{args.optimizer}'
    torch.manual_seed(args.seed)                             # before the model is built (train_rgb_depth.py:168)
    model = create_rgb_depth_model(base_channels=args.base_channels, bilinear=args.bilinear,
                                   output_size=cfg.dataset.images_size, max_depth=cfg.dataset.max_depth)
    di = _dist()
    trainer = FusedTrainer(_engine_of(model, args), 'DepthLoss', 1.0, 0.1, optimizer=args.optimizer, lr=args.learning_rate,
                           weight_decay=args.weight_decay, clip_norm=None, ddp=di[3])
    step = lambda tr, b: tr.step(b[0], b[1])[0]
    return _run(args, cfg, model, trainer, 'rgb', step, lambda m, b: m(b[0]), exp, dist_info=di)


# ---- train_binaural_attention.py ---------------------------------------------------------------------------------------
def main_binaural(argv=None):
    """/root/reference/train_binaural_attention.py:75-600: BinauralAttentionDepthNet, masked (gt > 0) L1 / SIlog /
    Combined loss (:240-290, 399-422), AdamW(lr 1e-3, wd 0.01), cosine schedule, no clipping."""
    from .models.binaural_attention_model import create_binaural_attention_model
    p = argparse.ArgumentParser(description='Train binaural attention depth model on Batvision dataset (MI355X)')
    _common_flags(p, 1e-3, 64)
    p.add_argument('--attention_levels', type=int, nargs='+', default=[2, 3, 4, 5])
    p.add_argument('--criterion', type=str, default='L1', choices=['L1', 'SIlog', 'Combined'])
    p.add_argument('--use_silog', type=lambda x: (str(x).lower() == 'true'), default=None)
    p.add_argument('--silog_lambda', type=float, default=0.5)
    p.add_argument('--l1_weight', type=float, default=0.5)
    p.add_argument('--silog_weight', type=float, default=0.5)
    args = p.parse_args(argv)
    cfg = load_config(dataset_name=args.dataset, model_name='unet_baseline', mode='train', experiment_name=args.experiment_name)
    exp = args.experiment_name or (f'binaural_attn_{args.dataset}_BS{args.batch_size}_Lr{args.learning_rate}_'
                                   f'{args.optimizer}_{args.criterion}')
    crit, l1w, sw = args.criterion, args.l1_weight, args.silog_weight
    if crit == 'Combined':                                   # reference :263-290
        use_silog = args.use_silog if args.use_silog is not None else (sw != 0.0)
        if not use_silog:
            sw = 0.0
    torch.manual_seed(args.seed)                             # before the model is built (train_binaural_attention.py:154)
    model = create_binaural_attention_model(base_channels=args.base_channels, bilinear=args.bilinear,
                                            output_size=cfg.dataset.images_size, max_depth=cfg.dataset.max_depth,
                                            attention_levels=args.attention_levels)
    if cfg.dataset.depth_norm:
        raise NotImplementedError('depth_norm with the binaural model (the head already outputs metres, reference :322-337)')
    di = _dist()
    trainer = FusedTrainer(_engine_of(model, args), crit, l1w, sw, args.silog_lambda, max_depth=cfg.dataset.max_depth,
                           optimizer=args.optimizer, lr=args.learning_rate, weight_decay=args.weight_decay, clip_norm=None,
                           mask_mode='gt0', ddp=di[3])
    step = lambda tr, b: tr.step(b[0], b[1])[0]
    return _run(args, cfg, model, trainer, 'audio', step, lambda m, b: m(b[0]), exp, dist_info=di)


# ---- train_adabins_distillation.py -----------------------------------------------------------------------------------------
def main_adabins(argv=None):
    """/root/reference/train_adabins_distillation.py:153-595: AdaBinsDistillationModel, DistillationLoss /
    AdaptiveDistillationLoss (:352-366), AdamW on the trainable parameters, CosineAnnealingLR(eta_min = lr / 100),
    clip_grad_norm_(1.0); checkpoints under ./results/<experiment>/ (best by validation RMSE, every 10 epochs)."""
    from .adabins_engine import AdaBinsTrainer
    from .models.adabins_distillation_model import create_adabins_distillation_model
    from .utils_distillation_loss import AdaptiveDistillationLoss, DistillationLoss
    p = argparse.ArgumentParser(description='Train AdaBins with Knowledge Distillation (MI355X)')
    p.add_argument('--dataset', type=str, default='batvisionv2', choices=['batvisionv1', 'batvisionv2'])
    p.add_argument('--n_bins', type=int, default=128)
    p.add_argument('--base_channels', type=int, default=64)
    p.add_argument('--max_depth', type=float, default=None)
    p.add_argument('--batch_size', type=int, default=None)
    p.add_argument('--learning_rate', '--lr', type=float, default=None)
    p.add_argument('--nb_epochs', type=int, default=None)
    p.add_argument('--optimizer', type=str, default='AdamW', choices=['Adam', 'AdamW', 'SGD'])
    p.add_argument('--use_adaptive_loss', action='store_true', default=False)
    p.add_argument('--freeze_rgb', action='store_true', default=False)
    p.add_argument('--temperature', type=float, default=4.0)
    p.add_argument('--lambda_task', type=float, default=1.0)
    p.add_argument('--lambda_response', type=float, default=0.5)
    p.add_argument('--lambda_feature', type=float, default=0.3)
    p.add_argument('--lambda_bin', type=float, default=0.2)
    p.add_argument('--lambda_sparse', type=float, default=0.1)
    p.add_argument('--checkpoints', type=int, default=0)
    p.add_argument('--experiment_name', type=str, default=None)
    p.add_argument('--use_wandb', action='store_true', default=False)
    p.add_argument('--wandb_project', type=str, default='batvision-depth-estimation')
    p.add_argument('--wandb_entity', type=str, default='branden')
    p.add_argument('--gpu_ids', type=str, default='0')
    p.add_argument('--synthetic', type=int, default=0)
    p.add_argument('--precision', type=str, default='bf16', choices=['bf16', 'f32', 'mxfp8'])
    args = p.parse_args(argv)
    cfg = load_config(dataset_name=args.dataset, model_name='unet_baseline', mode='train', experiment_name=args.experiment_name)
    if args.max_depth is not None:
        cfg.dataset.max_depth = args.max_depth
    args.batch_size = args.batch_size or cfg.mode.batch_size
    args.learning_rate = args.learning_rate or cfg.mode.learning_rate
    args.nb_epochs = args.nb_epochs or cfg.mode.epochs
    args.scheduler, args.eta_min, args.save_frequency = 'cosine', args.learning_rate * 0.01, 10
    args.num_workers, args.seed, args.device = cfg.mode.num_threads, 42, 'cuda'
    exp = args.experiment_name or f'adabins_distill_{args.dataset}_BS{args.batch_size}_Lr{args.learning_rate}_{args.optimizer}'
    torch.manual_seed(args.seed)
    model = create_adabins_distillation_model(n_bins=args.n_bins, base_channels=args.base_channels,
                                              output_size=cfg.dataset.images_size, max_depth=cfg.dataset.max_depth)
    if args.freeze_rgb:
        model.freeze_rgb()
    if args.use_adaptive_loss:
        criterion = AdaptiveDistillationLoss(max_epochs=args.nb_epochs, temperature=args.temperature,
                                             lambda_sparse=args.lambda_sparse)
    else:
        criterion = DistillationLoss(args.lambda_task, args.lambda_response, args.lambda_feature, args.lambda_bin,
                                     args.lambda_sparse, args.temperature)
    kind = 'audio' if (cfg.dataset.name == 'batvisionv1' and not args.synthetic) else 'both'
    di = _dist()
    trainer = AdaBinsTrainer.from_criterion(_engine_of(model, args), criterion, optimizer=args.optimizer, lr=args.learning_rate,
                                            clip_norm=1.0, ddp=di[3])
    def on_epoch(epoch):                                     # criterion.set_epoch(epoch) at every epoch start (:437-438)
        if args.use_adaptive_loss:
            criterion.set_epoch(epoch)
            trainer.set_criterion(criterion)

    def step(tr, b):
        audio, rgb, gt = (b[0], b[1], b[2]) if len(b) == 3 else (b[0], None, b[1])
        return tr.step(audio, rgb, gt)[0]

    fwd = lambda m, b: m(b[0], rgb=None, mode='inference')['audio']['final_depth']
    return _run(args, cfg, model, trainer, kind, step, fwd, exp, ckpt_root='results', ckpt_fmt='checkpoint_epoch_{:04d}.pth',
                on_epoch=on_epoch, dist_info=di)


# ---- train_base_residual.py ------------------------------------------------------------------------------------------------
def main_base_residual(argv=None):
    """/root/reference/train_base_residual.py:115-520: BaseResidualDepthNet, BaseResidualLoss / AdaptiveBaseResidualLoss
    (:258-281; --use_silog is store_true with default True, i.e. SIlog reconstruction), optimizer from the config,
    clip_grad_norm_(1.0) (:386), checkpoints under ./checkpoints/<experiment>/."""
    from .base_residual_engine import BaseResidualTrainer
    from .models.base_residual_model import create_base_residual_model
    from .utils_base_residual_loss import AdaptiveBaseResidualLoss, BaseResidualLoss
    p = argparse.ArgumentParser(description='Train Base+Residual depth model (MI355X)')
    p.add_argument('--dataset', type=str, default='batvisionv2', choices=['batvisionv1', 'batvisionv2'])
    p.add_argument('--audio_format', type=str, default='mel_spectrogram')
    p.add_argument('--base_channels', type=int, default=64)
    p.add_argument('--bilinear', action='store_true', default=True)
    p.add_argument('--use_adaptive_loss', action='store_true', default=False)
    p.add_argument('--use_silog', action='store_true', default=True)
    p.add_argument('--silog_lambda', type=float, default=0.5)
    p.add_argument('--lambda_recon', type=float, default=1.0)
    p.add_argument('--lambda_base', type=float, default=1.2)
    p.add_argument('--lambda_sparse', type=float, default=0.05)
    p.add_argument('--lowpass_kernel', type=int, default=16)
    p.add_argument('--warmup_epochs', type=int, default=50)
    p.add_argument('--batch_size', type=int, default=None)
    p.add_argument('--learning_rate', '--lr', type=float, default=None)
    p.add_argument('--optimizer', type=str, default=None, choices=['Adam', 'AdamW', 'SGD'])
    p.add_argument('--epochs', type=int, default=None)
    p.add_argument('--validation', type=lambda x: (str(x).lower() == 'true'), default=None)
    p.add_argument('--validation_iter', type=int, default=None)
    p.add_argument('--use_wandb', action='store_true', default=False)
    p.add_argument('--wandb_project', type=str, default='batvision-depth-estimation')
    p.add_argument('--wandb_entity', type=str, default='branden')
    p.add_argument('--experiment_name', type=str, default='base_res_default')
    p.add_argument('--checkpoints', type=int, default=None)
    p.add_argument('--synthetic', type=int, default=0)
    p.add_argument('--precision', type=str, default='bf16', choices=['bf16', 'f32', 'mxfp8'])
    args = p.parse_args(argv)
    cfg = load_config(dataset_name=args.dataset, model_name='unet_baseline', mode='train', experiment_name=args.experiment_name)
    cfg.dataset.audio_format = args.audio_format
    args.batch_size = args.batch_size or cfg.mode.batch_size
    args.learning_rate = args.learning_rate or cfg.mode.learning_rate
    args.nb_epochs = args.epochs or cfg.mode.epochs
    opt = args.optimizer or cfg.mode.optimizer
    args.scheduler, args.save_frequency = 'none', cfg.mode.saving_checkpoints
    args.num_workers, args.seed, args.device = cfg.mode.num_threads, 42, 'cuda'
    torch.manual_seed(args.seed)
    model = create_base_residual_model(input_channels=2, base_channels=args.base_channels, bilinear=args.bilinear,
                                       output_size=cfg.dataset.images_size, max_depth=cfg.dataset.max_depth)
    if args.use_adaptive_loss:
        criterion = AdaptiveBaseResidualLoss(lambda_recon_init=args.lambda_recon * 0.5,          # reference :261-269
                                             lambda_base_init=args.lambda_base * 2.0, lambda_sparse=args.lambda_sparse,
                                             warmup_epochs=args.warmup_epochs, lowpass_kernel=args.lowpass_kernel,
                                             use_silog=args.use_silog, silog_lambda=args.silog_lambda)
    else:
        criterion = BaseResidualLoss(lambda_recon=args.lambda_recon, lambda_base=args.lambda_base,
                                     lambda_sparse=args.lambda_sparse, lowpass_kernel=args.lowpass_kernel,
                                     use_silog=args.use_silog, silog_lambda=args.silog_lambda)
    di = _dist()
    trainer = BaseResidualTrainer.from_criterion(_engine_of(model, args), criterion, optimizer=opt, lr=args.learning_rate,
                                                 weight_decay=0.01 if opt == 'AdamW' else 0.0, clip_norm=1.0, ddp=di[3])

    def on_epoch(epoch):
        if args.use_adaptive_loss:
            criterion.set_epoch(epoch)
            trainer.set_criterion(criterion)

    step = lambda tr, b: tr.step(b[0], b[1])[0]
    return _run(args, cfg, model, trainer, 'audio', step, lambda m, b: m(b[0])[2], args.experiment_name,
                ckpt_fmt='checkpoint_{}.pth', on_epoch=on_epoch, dist_info=di)
