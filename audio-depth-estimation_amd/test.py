"""Evaluation entry point on libadn (mirror of the reference's test.py :24-366).

Same flags (:25-42), checkpoint resolution (--checkpoint_path | --experiment_name + --checkpoints under
./checkpoints/<exp>/checkpoint_<epoch>.pth, ``checkpoint["state_dict"]``), L1 test loss on ``gt != 0``,
per-sample ``compute_errors`` on metres (x max_depth when depth_norm, negatives clipped to 0), the seven mean
metrics, and the stats dict saved under ./eval/<dataset>/<split>/stats_on_<dataset>_<split>_set_<exp>_epoch_<n>.pt.
Forward pass, loss and metrics run on the device (one metrics kernel per batch instead of a numpy loop).
PNG visualisation is reporting-only and not provided: --visualize / --vis_batch_size are accepted (the reference's command
lines keep working) and a note is printed.  Extra flags: --synthetic N, --precision, --batch_size.
"""
import argparse
import os

import torch
from torch.utils.data import DataLoader

from .config_loader import load_config
from .dataloader.utils_dataset import GpuAudioFrontend
from .models.unetbaseline_model import *          # noqa: F401,F403
from .utils_criterion import compute_errors_batch
from .utils_loss import MaskedDepthLoss


def build_parser():
    p = argparse.ArgumentParser(description='Test U-Net model on Batvision dataset (MI355X)')
    p.add_argument('--dataset', type=str, default='batvisionv2', choices=['batvisionv1', 'batvisionv2'])
    p.add_argument('--experiment_name', type=str, default=None)
    p.add_argument('--checkpoint_path', type=str, default=None)
    p.add_argument('--checkpoints', type=int, default=50)
    p.add_argument('--eval_on', type=str, default='test', choices=['test', 'val'])
    p.add_argument('--visualize', action='store_true', default=False,
                   help='accepted for command-line compatibility (reference test.py:36); PNG panels are not produced')
    p.add_argument('--output_dir', type=str, default='./val/')
    p.add_argument('--vis_batch_size', type=int, default=4, help='accepted and ignored (reference test.py:40)')
    p.add_argument('--precision', default='bf16', choices=['bf16', 'f32'])
    p.add_argument('--synthetic', type=int, default=0)
    p.add_argument('--batch_size', type=int, default=None)
    return p


def resolve_checkpoint(args, cfg):
    if args.checkpoint_path is not None:
        if not os.path.exists(args.checkpoint_path):
            raise FileNotFoundError(f'Checkpoint not found: {args.checkpoint_path}')
        parts = args.checkpoint_path.split('/')
        name = next((q for q in parts if q.startswith('checkpoint_') and q.endswith('.pth')), None)
        epoch = int(name.replace('checkpoint_', '').replace('.pth', '')) if name else 0
        exp = parts[parts.index(name) - 1] if name and parts.index(name) > 0 else 'unknown'
        return args.checkpoint_path, exp, epoch
    if cfg.mode.checkpoints is None:
        raise AttributeError('In test mode, a checkpoint needs to be loaded. Provide --checkpoint_path or '
                             '--checkpoints with --experiment_name.')
    exp = args.experiment_name or cfg.mode.experiment_name or 'default'
    path = f'./checkpoints/{exp}/checkpoint_{cfg.mode.checkpoints}.pth'
    if not os.path.exists(path):
        raise FileNotFoundError(f'Checkpoint not found: {path}')
    return path, exp, cfg.mode.checkpoints


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.visualize:
        print('--visualize: PNG panels are not produced by this build (metrics and the stats file are); flag ignored')
    cfg = load_config(dataset_name=args.dataset, mode='test', experiment_name=args.experiment_name or 'default')
    if args.checkpoints is not None:
        cfg.mode.checkpoints = args.checkpoints
    cfg.mode.eval_on = args.eval_on
    if args.batch_size is not None:
        cfg.mode.batch_size = args.batch_size
    if cfg.mode.mode != 'test':
        raise Exception('This script is for test only. Please run train.py for training')
    if not torch.cuda.is_available():
        raise RuntimeError('test.py runs on libadn HIP kernels: no HIP device is visible (there is no CPU path)')
    device = torch.device('cuda', 0)
    S = cfg.dataset.images_size
    if args.synthetic:
        from .train import SyntheticBatvision
        eval_set, fe = SyntheticBatvision(args.synthetic, S, cfg.dataset.max_depth, cfg.dataset.depth_norm), None
    elif cfg.dataset.name == 'batvisionv1':
        from .dataloader.BatvisionV1_Dataset import BatvisionV1Dataset
        ann = cfg.dataset.annotation_file_val if args.eval_on == 'val' else cfg.dataset.annotation_file_test
        eval_set, fe = BatvisionV1Dataset(cfg, ann, frontend='raw'), GpuAudioFrontend('bv1', S)
    else:
        from .dataloader.BatvisionV2_Dataset import BatvisionV2Dataset
        ann = cfg.dataset.annotation_file_val if args.eval_on == 'val' else cfg.dataset.annotation_file_test
        eval_set = BatvisionV2Dataset(cfg, ann, frontend='raw')
        fe = GpuAudioFrontend(GpuAudioFrontend.bv2_mode(cfg.dataset.audio_format, cfg.dataset.max_depth), S)
    print(f'Eval Dataset of {len(eval_set)} instances')
    loader = DataLoader(eval_set, batch_size=cfg.mode.batch_size, shuffle=False, num_workers=cfg.mode.num_threads)

    model = define_G(cfg, input_nc=2, output_nc=1, ngf=64, netG=cfg.model.generator, norm='batch', use_dropout=False,  # noqa: F405
                     init_type='normal', init_gain=0.02, gpu_ids=[])
    model.compute_dtype = torch.bfloat16 if args.precision == 'bf16' else torch.float32
    path, exp, epoch = resolve_checkpoint(args, cfg)
    print(f'Loading checkpoint: {path}')
    ck = torch.load(path, map_location='cpu')
    model.load_state_dict({k[7:] if k.startswith('module.') else k: v for k, v in ck['state_dict'].items()})
    model = model.to(device).eval()
    l1 = MaskedDepthLoss('L1', mask_mode='ne0')
    md = float(cfg.dataset.max_depth)
    losses, rows, gts, preds = [], [], [], []
    with torch.no_grad():
        for audio, gt in loader:
            audio, gt = audio.to(device), gt.to(device)
            if fe is not None:
                audio = fe(audio)
            pred = model(audio)
            losses.append(l1(pred, gt).detach().reshape(1))
            sc = md if cfg.dataset.depth_norm else 1.0
            rows.append(compute_errors_batch((gt * sc).clamp(min=0.0), (pred * sc).clamp(min=0.0)))
            gts.append(gt[:, 0].cpu())
            preds.append(pred[:, 0].cpu())
    m = torch.cat(rows)
    mean = m.mean(0).tolist()
    print('\n' + '=' * 50 + '\nEvaluation Results:\n' + '=' * 50)
    for name, v in zip(('abs rel', 'RMSE', 'Delta1', 'Delta2', 'Delta3', 'Log10', 'MAE'), mean):
        print('{}: {:.3f}'.format(name, v))
    m = m.cpu()
    stats = {'loss': torch.cat(losses).cpu(), 'abs_rel': m[:, 0], 'rmse': m[:, 1], 'delta1': m[:, 2],
             'delta2': m[:, 3], 'delta3': m[:, 4], 'log10': m[:, 5], 'mae': m[:, 6],
             'gt_images': torch.cat(gts), 'pred_imgs': torch.cat(preds)}
    split = 'test' if cfg.mode.eval_on == 'test' else 'val'
    out_dir = os.path.join(cfg.mode.stat_dir, cfg.dataset.name, split)
    os.makedirs(out_dir, exist_ok=True)
    out_file = os.path.join(out_dir, f'stats_on_{cfg.dataset.name}_{split}_set_{exp}_epoch_{epoch}.pt')
    torch.save(stats, out_file)
    print(f'Evaluation results saved to: {out_file}')
    return mean


if __name__ == '__main__':
    main()
