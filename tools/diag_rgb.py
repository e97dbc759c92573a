"""Per-tensor gradient error of the RGB engine vs the float64 oracle (diagnostic, not a test)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_dcnet import _rgb, _oracle_step, max_rel, rel_l1
DEV = 'cuda'
bc = int(os.environ.get('BC', '64')); S = int(os.environ.get('S', '64')); B = int(os.environ.get('B', '2'))
for dtype in (torch.float32, torch.bfloat16):
    torch.manual_seed(0)
    model = _rgb(bc, S, dtype)
    with torch.no_grad():
        model.outc.bias.fill_(2.0)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    image = torch.rand(B, 3, S, S, generator=g)
    gt = 30 * torch.rand(B, 1, S, S, generator=g)
    from oracle import dcnet_oracle
    dcnet_oracle.QUANT = (lambda t: t.float().bfloat16().to(t.dtype)) if (dtype == torch.bfloat16 and os.environ.get('EMU', '1') == '1') else None
    pred_ref, loss_ref, grads_ref, stats_ref, pg = _oracle_step(sd, image, gt, 30.0)
    dcnet_oracle.QUANT = None
    model.train()
    eng = model.engine()
    pred = eng.forward(image.to(DEV), True).clone()
    print(dtype, 'pred rel_l1', rel_l1(pred, pred_ref))
    eng.backward(pg.float().to(DEV))
    for k, prm in model.named_parameters():
        got = eng.grad_view(prm).detach().float().cpu().reshape(-1)
        ref = grads_ref[k].reshape(-1).float()
        cos = float(torch.dot(got, ref) / (got.norm() * ref.norm() + 1e-30))
        print(f'  {k:45s} max_rel {max_rel(got, ref):.2e} cos {cos:.5f} |ref| {float(ref.abs().max()):.2e}')
