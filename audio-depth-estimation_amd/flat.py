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


def _align(n, a=8):       # 8 elements: 32-byte f32 slices, 16-byte slices of the bf16 mirror
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
        self._shape_sets = {}                       # shape key -> per-shape buffer set (see _shape_enter)
        self._packed_version = None
        self.weights_dirty = True
        self.s2_fresh = False
        self.on_grad_ready = None                   # callback(offset_lo): flat_g[offset_lo:] is final

    # ------------------------------------------------------------------ per-shape buffer sets
    # A captured hipGraph / launch plan holds raw device pointers into the engine's activation, workspace and
    # packed-weight buffers.  Those buffers are therefore never freed when another input shape comes along (a ragged
    # last validation batch between two graph replays): every shape key owns its own buffer set, kept alive in
    # ``_shape_sets`` and swapped in O(1).  Everything an engine assigns in its _prepare*() is per-shape state;
    # the names below are the shape-independent ones.
    _SHAPE_INDEPENDENT = frozenset((
        'module', 'dtype', 'requested_dtype', 'mx8', 'model_name', 'depth_norm', 'n', 'levels', '_build', '_saved', 'flat_p', 'flat_g', 'flat_w16',
        'param_meta', 'total', 'offset', 'on_grad_ready', '_shape_key', '_shape_sets', '_packed_version',
        'weights_dirty', 's2_fresh', 'train_offset', 'step_counter', 'dropout_seed'))
    MAX_SHAPE_SETS = 4

    def _shape_snapshot(self):
        snap = {k: v for k, v in self.__dict__.items() if k not in self._SHAPE_INDEPENDENT}
        if isinstance(getattr(self, 'levels', None), list):          # U-Net: per-level dicts are updated in place
            snap['__levels__'] = [dict(lv) for lv in self.levels]
        return snap

    def _shape_restore(self, snap):
        for k, v in snap.items():
            if k == '__levels__':
                for lv, saved in zip(self.levels, v):
                    lv.clear()
                    lv.update(saved)
            else:
                self.__dict__[k] = v

    def _shape_enter(self, key):
        """Make ``key`` the current shape.  Returns True when its buffer set was restored from the cache (nothing to
        build); False when the caller has to build it (the previous set has been parked, not freed)."""
        if key == self._shape_key:
            return True
        if self._shape_key is not None:
            self._shape_sets[self._shape_key] = self._shape_snapshot()
            while len(self._shape_sets) > self.MAX_SHAPE_SETS:       # oldest first; pinned sets are never dropped
                victim = next((k for k, v in self._shape_sets.items() if not v.get('_pinned')), None)
                if victim is None:
                    break
                del self._shape_sets[victim]
        snap = self._shape_sets.pop(key, None)
        if snap is None:
            self._pinned = False
            self._shape_key = None         # a build that raises leaves no half-built current set behind
            return False
        self._shape_restore(snap)
        self._shape_key = key
        self.weights_dirty = True          # the packed operands of this set date from its last use
        return True

    def pin_buffers(self):
        """Called by a graph / launch-plan capture: the current shape's buffer set must outlive the engine's use of it."""
        self._pinned = True

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
        self._shape_sets = {}              # views into the old flat buffers: rebuilt on demand

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

    def _mirror_fresh(self):
        """True when the bf16 parameter mirror written by the fused optimizer still matches the f32 masters: nobody
        (load_state_dict, an in-place edit) touched the parameters since the last pack."""
        return self.s2_fresh and self._packed_version == self._version_sum()
