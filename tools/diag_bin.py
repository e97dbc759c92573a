import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_dcnet import _binaural, GOLDEN, max_rel
from audio_depth_estimation_amd.engine import FusedTrainer
z = np.load(os.path.join(GOLDEN, 'binaural64_bc8.npz'))
bc, S, B = [int(v) for v in z['meta']]
lr, wd, max_depth, l1w, sw, lam = [float(v) for v in z['hyper']]
sd0 = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd0/')}
model = _binaural(bc, S, torch.float32, sd0, max_depth)
audio, gt = torch.from_numpy(z['audio']).cuda(), torch.from_numpy(z['gt']).cuda()
model.train()
tr = FusedTrainer(model.engine(), 'Combined', l1w, sw, lam, max_depth=max_depth, optimizer='AdamW', lr=lr, weight_decay=wd, clip_norm=None, mask_mode='gt0')
loss, pred = tr.step(audio, gt)
eng = model.engine()
for k, prm in model.named_parameters():
    if 'attention' in k or 'fusion' in k:
        got = eng.grad_view(prm).detach().float().cpu().reshape(-1)
        ref = torch.from_numpy(z['grad/' + k]).reshape(-1)
        print(f'{k:50s} {max_rel(got, ref):.2e}  got {got[:3].tolist()} ref {ref[:3].tolist()}')
