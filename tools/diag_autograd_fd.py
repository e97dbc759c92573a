"""Per-leaf finite-difference check of the AdaBins student's autograd bridge (diagnostic)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_depth_estimation_amd.models.adabins_distillation_model import AdaBinsDistillationModel

DEV = 'cuda'
g = torch.Generator().manual_seed(29)
audio = torch.rand(2, 2, 32, 32, generator=g).to(DEV)
torch.manual_seed(6)
m = AdaBinsDistillationModel(128, 64, 32, 30.0)
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
m.compute_dtype = torch.float32
m.freeze_rgb()
m = m.to(DEV).train()
names = ['x1', 'x2', 'x3', 'x4', 'x5', 'centers', 'logits', 'base', 'residual', 'final']
wts = {}


def leaves():
    o = m(audio, None, mode='train')['audio']
    return [o['features'][f'x{i}'] for i in range(1, 6)] + [o['bin_centers'], o['bin_logits'], o['base_depth'], o['residual'],
                                                            o['final_depth']]


def objective(i):
    t = leaves()[i]
    if i not in wts:
        wts[i] = torch.randn(t.shape, generator=torch.Generator().manual_seed(100 + i)).to(DEV) / t.numel() ** 0.5
    return (t * wts[i]).sum()


pname = sys.argv[1] if len(sys.argv) > 1 else 'audio_encoder.down2.maxpool_conv.1.double_conv.0.weight'
p = dict(m.named_parameters())[pname]
d = torch.randn(p.shape, generator=torch.Generator().manual_seed(7)).to(DEV)
d /= d.norm()
for i, n in enumerate(names):
    m.zero_grad()
    objective(i).backward()
    analytic = float((p.grad.double() * d.double()).sum()) if p.grad is not None else float('nan')
    row = [f'{n:9s} analytic {analytic:+.6f}']
    for h in (4e-2, 2e-2, 1e-2, 5e-3):
        with torch.no_grad():
            p.add_(h * d)
            up = float(objective(i))
            p.sub_(2 * h * d)
            dn = float(objective(i))
            p.add_(h * d)
        row.append(f'h={h:g}: {(up - dn) / (2 * h):+.6f}')
    print('   '.join(row), flush=True)
