"""A checkpoint WRITTEN BY THE REFERENCE's code path, plus what the reference computes when it resumes from it
(build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_checkpoint.py

Mirrors /root/reference/train_binaural_attention.py: model = create_binaural_attention_model(...) (:297-307),
optimizer = torch.optim.AdamW(model.parameters(), lr, weight_decay) (:320-325), the training step (:394-433: masked L1
over depth_gt > 0, zero_grad / backward / step), the checkpoint dict of :563-571 written with torch.save, and the resume
of :358-361 (model.load_state_dict + optimizer.load_state_dict) followed by one more step.

Outputs: ref_ckpt_binaural_bc4.pth (the checkpoint after 2 steps: pure tensor data, ~0.9 MB) and
ref_ckpt_binaural_bc4_next.npz (inputs, loss and a parameter sample of step 3 after the resume).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '/root/reference')
from models.binaural_attention_model import create_binaural_attention_model   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
LR, WD = 1e-3, 0.01


def batch(seed, B=2, S=64):
    g = torch.Generator().manual_seed(seed)
    audio = torch.rand(B, 2, S, S, generator=g)
    gt = 30 * torch.rand(B, 1, S, S, generator=g)
    gt[gt < 3] = 0
    return audio, gt


def step(model, opt, audio, gt):
    pred = model(audio)
    valid = gt > 0
    loss = torch.nn.L1Loss()(pred[valid], gt[valid])
    opt.zero_grad()
    loss.backward()
    opt.step()
    return float(loss)


def main():
    torch.set_num_threads(8)
    torch.manual_seed(42)
    model = create_binaural_attention_model(base_channels=4, bilinear=True, output_size=64, max_depth=30.0,
                                            attention_levels=[2, 3, 4, 5])
    with torch.no_grad():
        for m in model.attention_modules.values():
            m.gamma.fill_(0.3)                     # gamma = 0 would leave the attention path untested
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=LR, weight_decay=WD)
    losses = [step(model, opt, *batch(200 + i)) for i in range(2)]
    path = os.path.join(HERE, 'ref_ckpt_binaural_bc4.pth')
    torch.save({'epoch': 2, 'model_state_dict': model.state_dict(), 'optimizer_state_dict': opt.state_dict(),
                'train_loss': losses[-1], 'val_loss': 0.0, 'val_errors': {}}, path)
    # resume exactly as the reference does, then one more step
    ck = torch.load(path)
    torch.manual_seed(7)
    model2 = create_binaural_attention_model(base_channels=4, bilinear=True, output_size=64, max_depth=30.0,
                                             attention_levels=[2, 3, 4, 5])
    model2.train()
    opt2 = torch.optim.AdamW(model2.parameters(), lr=LR, weight_decay=WD)
    model2.load_state_dict(ck['model_state_dict'])
    opt2.load_state_dict(ck['optimizer_state_dict'])
    audio, gt = batch(300)
    before = {k: p.detach().clone() for k, p in model2.named_parameters()}
    l3 = step(model2, opt2, audio, gt)
    out = {'audio': audio.numpy(), 'gt': gt.numpy(), 'loss': np.float64(l3), 'hyper': np.array([LR, WD])}
    for k, p in model2.named_parameters():
        out['delta/' + k] = (p.detach() - before[k]).reshape(-1)[:256].numpy()
    np.savez_compressed(os.path.join(HERE, 'ref_ckpt_binaural_bc4_next.npz'), **out)
    print('checkpoint bytes', os.path.getsize(path), 'step-3 loss', l3)


if __name__ == '__main__':
    main()
