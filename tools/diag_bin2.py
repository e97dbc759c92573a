import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_dcnet import _binaural, rel_l1
from oracle import dcnet_oracle, loss_oracle
DEV='cuda'
for dtype in (torch.bfloat16,):
    torch.manual_seed(0)
    S = 64
    model = _binaural(64, S, dtype)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for m in model.attention_modules.values():
            m.gamma.fill_(0.5)
            for conv in (m.query, m.key, m.value, m.out):
                conv.bias.copy_(0.1 * torch.randn(conv.bias.shape, generator=g))
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    audio = torch.rand(2, 2, S, S, generator=g)
    gt = 30 * torch.rand(2, 1, S, S, generator=g); gt[gt < 3] = 0
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    pkeys = [k for k, v in sd64.items() if v.is_floating_point() and 'running_' not in k]
    for k in pkeys: sd64[k].requires_grad_(True)
    dcnet_oracle.QUANT = lambda t: t.float().bfloat16().to(t.dtype)
    pred_ref, _ = dcnet_oracle.binaural_forward(sd64, audio.double(), 30.0, training=True)
    dcnet_oracle.QUANT = None
    pred_ref.retain_grad()
    loss_oracle.masked_loss(pred_ref, gt.double(), 'L1', mask_mode='gt0').backward()
    model.train(); eng = model.engine()
    pred = eng.forward(audio.to(DEV), True).clone()
    print('pred rel_l1', rel_l1(pred, pred_ref.detach()))
    eng.backward(pred_ref.grad.float().to(DEV))
    for k, prm in model.named_parameters():
        got = eng.grad_view(prm).detach().float().cpu().reshape(-1); ref = sd64[k].grad.reshape(-1).float()
        cos = float(torch.dot(got, ref) / (got.norm() * ref.norm() + 1e-30))
        print(f'{k:55s} cos {cos:.4f}' + (f'  got {float(got[0]):.4e} ref {float(ref[0]):.4e}' if got.numel() == 1 else ''))
