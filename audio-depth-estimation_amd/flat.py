"""Flat f32 parameter / gradient buffers shared by the libadn engines.

All trainable parameters of a module live in ONE flat f32 buffer (parameters() order, each tensor 16-byte
aligned), conv weights in torch channels_last memory order ([Cout][kh][kw][Cin]) which is the row layout the
implicit-GEMM kernels read; the module's nn.Parameters are views into that buffer so state_dict() /
load_state_dict() / torch.optim keep working.  Gradients use a second flat buffer with the same offsets (the
bucket source of the data-parallel all-reduce, ddp.py) and, on the bf16 path, a bf16 mirror of the parameters
that the fused optimizer refreshes every step.
"""
from __future__ import annotations

import torch


def _align(n, a=4):
    return (n + a - 1) // a * a


class FlatParamEngine:
    """Parameter plumbing of an engine; subclasses set ``module``, ``dtype`` and ``model_name``."""
    model_name = 'model'

    def _init_flat(self):
        self.flat_p = None
        self.flat_g = None
        self.flat_w16 = None
        self.param_meta = []                        # (param, offset, numel)
        self._shape_key = None
        self._packed_version = None
        self.weights_dirty = True
        self.s2_fresh = False
        self.on_grad_ready = None                   # callback(offset_lo): flat_g[offset_lo:] is final

    def _bound(self):
        if self.flat_p is None:
            return False
        p0, off0, _ = self.param_meta[0]
        pl, offl, _ = self.param_meta[-1]
        base = self.flat_p.data_ptr()
        return p0.data_ptr() == base + 4 * off0 and pl.data_ptr() == base + 4 * offl

    def bind_parameters(self):
        """(Re)create the flat parameter/gradient buffers and re-point the module's Parameters into them."""
        params = list(self.module.parameters())
        dev = params[0].device
        if dev.type != 'cuda':
            raise RuntimeError(f'{self.model_name} runs on libadn HIP kernels only: move the model to a HIP device '
                               '(gpu_ids=[0] or .to("cuda")); there is no CPU path')
        total, meta = 0, []
        for p in params:
            meta.append((p, total, p.numel()))
            total += _align(p.numel())
        flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, off, n in meta:
            view = self._view(flat_p, off, p)
            view.copy_(p.data)
            p.data = view
            p.grad = None
        self.flat_p, self.flat_g, self.param_meta, self.total = flat_p, flat_g, meta, total
        # bf16 mirror of the parameters (same offsets): the fused optimizer writes it, the S2 GEMM operands
        # of the unpadded layers are views into it
        self.flat_w16 = torch.zeros(total, dtype=torch.bfloat16, device=dev) if self.dtype == torch.bfloat16 else None
        self.offset = {id(p): off for p, off, _ in meta}
        self.weights_dirty = True
        self.s2_fresh = False
        self._shape_key = None

    @staticmethod
    def _view(flat, off, p):
        n = p.numel()
        if p.dim() == 4:
            X, Y, kh, kw = p.shape
            return flat[off:off + n].view(X, kh, kw, Y).permute(0, 3, 1, 2)
        return flat[off:off + n].view(p.shape)

    def grad_view(self, p):
        return self._view(self.flat_g, self.offset[id(p)], p)

    def _flat_slice(self, buf, p):
        off = self.offset[id(p)]
        return buf[off:off + p.numel()]

    def _version_sum(self):
        return sum(p._version for p, _, _ in self.param_meta)
