"""Debug aid: ConvT2x2 op output vs torch.conv_transpose2d inside an RGBDepthNet(bilinear=False)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from audio_depth_estimation_amd.models.rgb_depth_model import RGBDepthNet

bc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
m = RGBDepthNet(bc, False, 64, 30.0)
m.compute_dtype = torch.float32
m = m.to('cuda').eval()
x = torch.rand(2, 3, 64, 64, device='cuda')
with torch.no_grad():
    m(x)
eng = m.engine()
acts = {a.name: a for a in eng.acts}
print(sorted(acts))
for i, (src, dst) in enumerate((('x5', 'd4.up'), ('d4', 'd3.up'), ('d3', 'd2.up'), ('d2', 'd1.up'))):
    up = getattr(m, f'up{i + 1}').up
    s = acts[src]
    xin = s.data[..., :getattr(s, 'C_real', s.C)].permute(0, 3, 1, 2).float()
    want = F.conv_transpose2d(xin, up.weight, up.bias, stride=2)
    got = acts[dst].data.permute(0, 3, 1, 2).float()
    print(src, dst, tuple(xin.shape), tuple(got.shape), float((got - want).abs().max()), float(want.abs().max()))

op = [o for o in eng.ops if type(o).__name__ == 'ConvT2x2'][0]
up = m.up1.up
W = up.weight.detach()                      # [Cin][Cout][2][2]
cin, cout = W.shape[0], W.shape[1]
want_w = W.permute(2, 3, 1, 0).reshape(4 * cout, cin)        # [(t,co)][ci]
print('w_fwd', float((op.w_fwd[:, :cin].float() - want_w).abs().max()), tuple(op.w_fwd.shape), op.w_fwd.stride())
s = acts['x5']
xin = s.data.reshape(-1, cin).float()
want_tmp = xin @ want_w.t() + up.bias.detach().repeat(4)
print('tmp', float((op.tmp.reshape(-1, 4 * cout).float() - want_tmp).abs().max()), float(want_tmp.abs().max()))
print('bias4', float((op.bias4 - up.bias.detach().repeat(4)).abs().max()))
