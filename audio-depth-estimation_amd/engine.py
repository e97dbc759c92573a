"""Fused U-Net pipeline on libadn kernels: forward, backward, loss, clip, optimizer.

This is the MI355X-side replacement of what PyTorch dispatches for the reference's
``model(audio)`` / ``loss.backward()`` / ``clip_grad_norm_`` / ``optimizer.step()``
(/root/reference/train.py:642-691) for ``models.unetbaseline_model.UnetGenerator``.

Data layout in HBM
  * activations: NHWC, dtype f32 (exact path) or bf16 (throughput path); per down level i the raw conv
    output ``zd[i]`` (BN levels only) plus the two materialised consumer views ``ad[i]`` =
    LeakyReLU(BN(z)) (operand of the next down conv) and ``rd[i]`` = ReLU(BN(z)) (skip operand of the
    same level's transposed conv); per up level ``zu[i]`` and ``ru[i]`` = ReLU(BN(zu)).  The skip
    concat (unetbaseline_model.py:235) is virtual: the consumer GEMM reads two base pointers.
  * parameters / gradients / Adam moments: one flat f32 buffer each, parameters() order, conv weights
    in torch channels_last memory order ([X][4][4][Y]) which IS the S2 GEMM operand layout; the module's
    nn.Parameters are views into the flat buffer so state_dict()/load_state_dict()/torch.optim keep working.
  * per step the f32 master weights are cast/packed to the two GEMM operand forms (adn_pack_weights).
Gradients become final from the END of the flat buffer towards its start (outermost up layer first,
outermost down layer last), which is the bucket order of the data-parallel all-reduce (ddp.py).
"""
from __future__ import annotations

import os

import torch

from . import _lib
from . import kernels as K
from .flat import FlatParamEngine
from ._lib import EPI_ACT, EPI_BWD, EPI_FINAL, EPI_Z_STATS, GEMM_S2, GEMM_T2

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LEAKY = 0.2


class UNetEngine(FlatParamEngine):
    """Runs UnetGenerator.forward/backward through libadn.  One engine per module instance."""

    def __init__(self, module, num_downs, depth_norm, compute_dtype=torch.bfloat16):
        self.module = module
        self.n = num_downs
        self.depth_norm = bool(depth_norm)
        self.dtype = compute_dtype
        self.levels = module._adn_levels()          # list of dicts with layer objects, outermost first
        assert len(self.levels) == num_downs
        self.model_name = 'UnetGenerator'
        self._init_flat()
        self._saved = None

    def _pack_weights(self):
        """Cast/pack the f32 master weights into the GEMM operand forms.

        S2 operands: views of the bf16 mirror (written by the fused optimizer; re-cast here only when the
        parameters were changed from outside) or the f32 master itself; padded edge layers get their own
        small pack.  T2 operands: every layer in ONE launch (adn_pack_t2_multi).
        """
        T = self.dtype
        fresh = self._mirror_fresh()
        for lv in self.levels:
            for key in ('down', 'up'):
                w = lv[key].weight
                X, Y = w.shape[0], w.shape[1]
                master = self._flat_slice(self.flat_p, w)
                ypad = lv[key + '_ypad']
                if ypad != Y:
                    K.pack_weights(master, X, Y, T, lv[key + '_s2'], None, y_pad=ypad)
                elif T == torch.float32:
                    lv[key + '_s2'] = master                       # channels_last memory == S2 operand
                elif not fresh:
                    K.pack_weights(master, X, Y, T, lv[key + '_s2'], None)
        # source of the T2 pack: the bf16 mirror when the fused optimizer has just written ALL of it (half the read
        # traffic, identical values); the f32 masters otherwise (the re-cast above skips the padded edge layers)
        src = self.flat_w16 if (T == torch.bfloat16 and fresh) else self.flat_p
        K.pack_t2_multi(src, self.t2_table, self.t2_layers, self.t2_blocks, T, self.t2_all)
        self.weights_dirty = False
        self.s2_fresh = False
        self._packed_version = self._version_sum()

    # ------------------------------------------------------------------ buffers
    def _prepare(self, x):
        if not self._bound():
            self.bind_parameters()
        B, Cin, H, W = x.shape
        n = self.n
        if H % (1 << n) or W % (1 << n):
            raise RuntimeError(f'input {H}x{W} is not divisible by 2^{n}: Kernel size can\'t be greater than '
                               f'actual input size')
        key = (B, Cin, H, W, x.device)
        if self._shape_enter(key):
            return
        dev, T = x.device, self.dtype
        f32 = dict(dtype=torch.float32, device=dev)
        ws_bytes = 16
        epc = 8 if T == torch.bfloat16 else 4          # elements per 16-byte chunk
        l0 = self.levels[0]
        # Edge layers: the first conv (Cin = 2) and the last transposed conv (Cout = 1) run on the MFMA
        # kernels with their thin channel dimension zero-padded to one 16-byte chunk.
        self.cin_pad = Cin
        if l0['down'].weight.shape[0] % 64 == 0 and Cin % epc:
            self.cin_pad = (Cin + epc - 1) // epc * epc
        cu_in0, cout0 = l0['up'].weight.shape[0], l0['up'].weight.shape[1]
        ks = 4 * epc
        self.n1_path = (cout0 == 1 and cu_in0 % 64 == 0 and (cu_in0 // 2) % ks == 0)
        self.cout_pad = epc if self.n1_path else cout0
        # Dedicated HBM-bound kernels for the thin outermost layers (csrc/edge.hip; bf16 path, the unet_256 / ngf 64
        # geometry): first conv 2 -> 64 forward + weight gradient, last transposed conv 128 -> 1 input / weight gradient.
        # They read the thin operand (network input, output gradient) as planar f32: no channel padding at all.
        self.edge_path = (T == torch.bfloat16 and self.n1_path and n >= 2 and Cin == 2
                          and l0['down'].weight.shape[0] == 64 and cu_in0 == 128 and l0['bn_d'] is None
                          and (W // 2) % 32 == 0 and not os.environ.get('ADN_NO_EDGE'))
        if self.edge_path:
            self.cin_pad, self.cout_pad = Cin, cout0
        self.x_nhwc = None if self.edge_path else torch.empty(B, H, W, self.cin_pad, dtype=T, device=dev)
        batching_patch = (T == torch.bfloat16 and not os.environ.get('ADN_NO_WGRAD_BATCH')
                          and not os.environ.get('ADN_NO_PATCH_BATCH'))
        pshape_d, pshape_u = {}, {}
        for i, lv in enumerate(self.levels):
            dw, uw = lv['down'].weight, lv['up'].weight
            cd_in, cd_out = dw.shape[1], dw.shape[0]
            cu_in, cu_out = uw.shape[0], uw.shape[1]
            hs, wsz = H >> (i + 1), W >> (i + 1)
            lv.update(hs=hs, ws=wsz, cd_in=cd_in, cd_out=cd_out, cu_in=cu_in, cu_out=cu_out)
            cd_in_p = self.cin_pad if i == 0 else cd_in          # gathered channels as the kernels see them
            cu_out_p = self.cout_pad if i == 0 else cu_out
            lv.update(down_ypad=cd_in_p, up_ypad=cu_out_p)
            act = lambda c, h=hs, w_=wsz: torch.empty(B, h, w_, c, dtype=T, device=dev)
            lv['ad'] = act(cd_out) if i < n - 1 else None
            lv['rd'] = act(cd_out)
            lv['Gd'] = act(cd_out)
            lv['zd'] = act(cd_out) if lv['bn_d'] is not None else None
            big = lambda c, h=hs, w_=wsz: torch.empty(B, 2 * h, 2 * w_, c, dtype=T, device=dev)
            if i > 0:
                lv['zu'] = big(cu_out)
                lv['ru'] = big(cu_out)
                lv['Gu'] = big(cu_out)
            else:
                lv['out'] = torch.empty(B, 2 * hs, 2 * wsz, cu_out, **f32)
                lv['dz0'] = torch.empty(B, 1, 2 * hs, 2 * wsz, **f32) if self.edge_path else big(cu_out_p)
            # packed weights: S2 = view of the bf16 parameter mirror (unpadded bf16 layers), own buffer when
            # padded, the f32 master itself on the exact path; T2 = slices of one flat buffer (below)
            for wk, X, Y, Yp in (('down', cd_out, cd_in, cd_in_p), ('up', cu_in, cu_out, cu_out_p)):
                if Yp != Y:
                    lv[wk + '_s2'] = torch.empty(X, 16, Yp, dtype=T, device=dev)
                elif T != torch.float32:
                    lv[wk + '_s2'] = self._flat_slice(self.flat_w16, lv[wk].weight).view(X, 16, Y)
            # GEMM plans: partial rows + workspace
            c_up0 = cd_out
            c_up1 = cu_in - cd_out
            edge0 = i == 0 and self.edge_path
            pd, w1 = (0, 0) if edge0 else K.igemm_query(T, GEMM_S2, B, hs, wsz, cd_in_p, 0, cd_out, [cd_out],   # Li fwd
                                                       epi=EPI_Z_STATS if lv['bn_d'] is not None else EPI_ACT)
            if i == 0 and self.n1_path:
                pu, w2 = 0, K.convt_n1_workspace_bytes(B, hs, wsz)                                  # D0 fwd
            else:
                pu, w2 = K.igemm_query(T, GEMM_T2, B, hs, wsz, c_up0, c_up1, cu_out, [cu_out],     # Di fwd
                                        epi=EPI_Z_STATS if lv['bn_u'] is not None else EPI_FINAL)
            pgu, w3 = (0, 0) if edge0 else K.igemm_query(T, GEMM_S2, B, hs, wsz, cu_out_p, 0, cu_in,   # Di dgrad
                                                         [c_up0, c_up1] if c_up1 else [c_up0], epi=EPI_BWD)
            pgd, w4 = (0, 0) if i == 0 else K.igemm_query(T, GEMM_T2, B, hs, wsz, cd_out, 0, cd_in, [cd_in], epi=EPI_BWD)  # Li dgrad
            w5 = 0 if edge0 else K.wgrad_workspace_bytes(T, B, hs, wsz, cd_out, 0, cd_in_p, 0,
                                                         cd_in if cd_in_p != cd_in else 0)
            w6 = 0 if edge0 else K.wgrad_workspace_bytes(T, B, hs, wsz, c_up0, c_up1, cu_out_p, 0,
                                                         cu_out if cu_out_p != cu_out else 0)
            if i == 0 and self.edge_path:
                pgu = K.d0_dgrad_num_partials(B, hs, wsz)
                w3 = 0
                w5 = K.thin_wgrad_workspace_bytes(B, hs, wsz, 2, 64, 0)
                w6 = K.thin_wgrad_workspace_bytes(B, hs, wsz, 1, 64, 64)
            ws_bytes = max(ws_bytes, w1, w2, w3, w4, w5, w6)
            lv.update(P_d=pd, P_u=pu, P_gu=pgu, P_gd=pgd)
            # small-image levels: their weight gradients ride in one multi-problem launch (adn_wgrad_batch)
            # patch-staged levels: candidates for the shared-split launch (see _patch_groups below)
            if batching_patch and not edge0:
                if cd_in_p == cd_in and K.wgrad_patch_batch_workspace_bytes(T, B, [(hs, wsz, cd_out, 0, cd_in_p, 0)]) >= 0:
                    pshape_d[i] = (hs, wsz, cd_out, 0, cd_in_p, 0)
                if cu_out_p == cu_out and K.wgrad_patch_batch_workspace_bytes(T, B, [(hs, wsz, c_up0, c_up1, cu_out_p, 0)]) >= 0:
                    pshape_u[i] = (hs, wsz, c_up0, c_up1, cu_out_p, 0)
            # (class of the problem -- a launch holds one class -- and its number of norm partials; 0 = own launch)
            batching = T == torch.bfloat16 and not os.environ.get('ADN_NO_WGRAD_BATCH')
            lv['wb_d'], lv['wbq_d'] = (K.wgrad_batchable(T, B, hs, wsz, cd_out, 0, cd_in_p, 0)
                                       if batching and not edge0 and cd_in_p == cd_in else (0, 0))
            lv['wb_u'], lv['wbq_u'] = (K.wgrad_batchable(T, B, hs, wsz, c_up0, c_up1, cu_out_p, 0)
                                       if batching and not edge0 and cu_out_p == cu_out else (0, 0))
            for tag, bn, C in (('d', lv['bn_d'], cd_out), ('u', lv['bn_u'], cu_out)):
                if bn is None:
                    continue
                for nm in ('mean', 'istd', 'scale', 'shift'):
                    lv[f'{nm}_{tag}'] = torch.empty(C, **f32)
                lv[f'coef_{tag}'] = torch.empty(2 * C, **f32)
                lv[f'part_{tag}'] = torch.empty((pd if tag == 'd' else pu) * 2 * C, **f32)
        # backward stats partials: the tensor BN'd at (level i, tag) receives its final gradient from
        #   tag 'u' (zu[i], i>=1): dgrad of D(i-1)  -> P_gu of level i-1
        #   tag 'd' (zd[i], 1<=i<=n-2): dgrad of L(i+1) -> P_gd of level i+1
        for i, lv in enumerate(self.levels):
            if lv['bn_u'] is not None:
                lv['bpart_u'] = torch.empty(self.levels[i - 1]['P_gu'] * 2 * lv['cu_out'], **f32)
            if lv['bn_d'] is not None:
                lv['bpart_d'] = torch.empty(self.levels[i + 1]['P_gd'] * 2 * lv['cd_out'], **f32)
        # one flat T2 buffer + the layer table of adn_pack_t2_multi
        rows, t2_off, blk = [], 0, 0
        for lv in self.levels:
            for wk in ('down', 'up'):
                w = lv[wk].weight
                X, Y = w.shape[0], w.shape[1]
                rows.append([self.offset[id(w)], X, Y, t2_off, blk])
                lv[wk + '_t2_off'] = t2_off
                t2_off += 16 * X * Y
                blk += ((X + 63) // 64) * ((Y + 63) // 64) * 16
        self.t2_all = torch.empty(t2_off, dtype=T, device=dev)
        for lv in self.levels:
            for wk in ('down', 'up'):
                w = lv[wk].weight
                o = lv[wk + '_t2_off']
                lv[wk + '_t2'] = self.t2_all[o:o + 16 * w.shape[0] * w.shape[1]]
        # The patch-staged weight gradients of consecutive wide levels share ONE launch with 1/n of the pixel splits each
        # (adn_wgrad_patch_batch: n layers then write and re-read 1/n of the f32 slabs each): the down group L3, L2, L1 goes
        # out when L1's output gradient is final, the up group D1, D2, D3 when D3's is.
        def group(shapes):
            idx = sorted(shapes)[:4]
            ok = len(idx) >= 2 and idx == list(range(idx[0], idx[0] + len(idx)))
            return idx if ok else []
        self.pb_d, self.pb_u = group(pshape_d), group(pshape_u)
        self.pshape = {('d', i): pshape_d[i] for i in self.pb_d}
        self.pshape.update({('u', i): pshape_u[i] for i in self.pb_u})
        for tag, idx in (('d', self.pb_d), ('u', self.pb_u)):
            if idx:
                ws_bytes = max(ws_bytes, K.wgrad_patch_batch_workspace_bytes(T, B, [self.pshape[(tag, i)] for i in idx]))
        self._prepare_fused_norm(B, dev)
        self.t2_table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self.t2_layers, self.t2_blocks = len(rows), blk
        self.workspace = torch.empty(ws_bytes // 4 + 4, **f32)
        self.red_ws = torch.empty(4096, dtype=torch.float64, device=dev)
        self.weights_dirty = True
        self._shape_key = key
        self.B = B

    # ------------------------------------------------------------------ fused gradient norm
    supports_fused_norm = True
    NORM_ROW = 8192          # elements per row of adn_grad_norm_ranges

    def _prepare_fused_norm(self, B, dev):
        """clip_grad_norm_ (train.py:689) without a pass over the 54 M-element gradient: the kernel that writes a
        conv layer's final dW leaves partial sums of dW^2 in ``sq_all`` (AdnWgradDesc.sq_partials); the parameters
        without such a kernel (BatchNorm affine, bias, the thin edge layers) are covered by ``norm_ranges``."""
        T = self.dtype
        self.sq_all = self.norm_ranges = None
        if os.environ.get('ADN_NO_FUSED_NORM'):
            return
        counts, covered = [], set()
        for i, lv in enumerate(self.levels):
            edge0 = i == 0 and self.edge_path
            hs, wsz = lv['hs'], lv['ws']
            c_up0, c_up1 = lv['cd_out'], lv['cu_in'] - lv['cd_out']
            nd = 0 if edge0 else K.wgrad_sq_count(T, B, hs, wsz, lv['cd_out'], 0, lv['down_ypad'], 0,
                                                  lv['cd_in'] if lv['down_ypad'] != lv['cd_in'] else 0)
            nu = 0 if edge0 else K.wgrad_sq_count(T, B, hs, wsz, c_up0, c_up1, lv['up_ypad'], 0,
                                                  lv['cu_out'] if lv['up_ypad'] != lv['cu_out'] else 0)
            if lv['wb_d']:                   # a problem of the multi-problem launch runs unsplit: its own partial count
                nd = lv['wbq_d']
            if lv['wb_u']:
                nu = lv['wbq_u']
            for wk, cnt in (('down', nd), ('up', nu)):
                counts.append((lv, wk, cnt))
                if cnt:
                    covered.add(id(lv[wk].weight))
        rows = []
        for p, off, numel in self.param_meta:
            if id(p) in covered:
                continue
            for o in range(0, numel, self.NORM_ROW):
                rows.append([off + o, (min(self.NORM_ROW, numel - o) + 3) // 4 * 4])   # the tail reads zero padding
        if not covered or len(rows) > 1024:
            return
        self.sq_all = torch.zeros(sum(c for _, _, c in counts), dtype=torch.float64, device=dev)
        o = 0
        for lv, wk, cnt in counts:
            lv[wk + '_sq'] = self.sq_all[o:o + cnt] if cnt else None
            o += cnt
        self.norm_ranges = torch.tensor(rows, dtype=torch.int64, device=dev).view(-1, 2) if rows else None

    # ------------------------------------------------------------------ forward
    def forward(self, x, training):
        if not x.is_cuda:
            raise RuntimeError('UnetGenerator.forward needs a HIP device tensor (libadn has no CPU path)')
        x = x.contiguous().float()
        self._prepare(x)
        if self.weights_dirty or self._packed_version != self._version_sum():
            self._pack_weights()
        T, B, n, L, ws = self.dtype, self.B, self.n, self.levels, self.workspace
        if self.edge_path:
            self._x_in, self._x_ver = x, x._version          # the first conv and its weight gradient read x itself
        else:
            K.nchw_to_nhwc(x, self.x_nhwc)
        # ---- down path
        for i, lv in enumerate(L):
            src = self.x_nhwc if i == 0 else L[i - 1]['ad']
            C, hs, wsz = lv['cd_out'], lv['hs'], lv['ws']
            bn = lv['bn_d']
            if i == 0 and self.edge_path:
                K.l0_forward(x, self._flat_slice(self.flat_p, lv['down'].weight), B, hs, wsz, LEAKY, lv['ad'], lv['rd'])
            elif bn is None:
                K.igemm(T, GEMM_S2, B, hs, wsz, src, None, lv['down_s2'], C, EPI_ACT,
                        [K.Seg(C, out0=lv['ad'], out1=lv['rd'], slope=LEAKY)], ws, algo_c=lv['cd_in'])
            elif training:
                K.igemm(T, GEMM_S2, B, hs, wsz, src, None, lv['down_s2'], C, EPI_Z_STATS,
                        [K.Seg(C, out0=lv['zd'], partials=lv['part_d'])], ws)
                self._bn_forward(lv, 'd', bn, B * hs * wsz, lv['P_d'], lv['zd'], LEAKY, lv['ad'], lv['rd'])
            else:
                K.bn_eval_affine(bn.weight, bn.bias, bn.running_mean, bn.running_var, BN_EPS, lv['scale_d'],
                                 lv['shift_d'])
                K.igemm(T, GEMM_S2, B, hs, wsz, src, None, lv['down_s2'], C, EPI_ACT,
                        [K.Seg(C, out0=lv['ad'], out1=lv['rd'], scale=lv['scale_d'], shift=lv['shift_d'],
                               slope=LEAKY)], ws)
        # ---- up path
        for i in reversed(range(n)):
            lv = L[i]
            in0 = lv['rd']
            in1 = L[i + 1]['ru'] if i < n - 1 else None
            C, hs, wsz = lv['cu_out'], lv['hs'], lv['ws']
            bn = lv['bn_u']
            if i == 0:
                bias = lv['up'].bias
                fa = 1 if self.depth_norm else 0
                if self.n1_path:
                    K.convt_n1_forward(T, B, hs, wsz, in0, in1, self._flat_slice(self.flat_p, lv['up'].weight), bias,
                                       fa, lv['out'], ws)
                else:
                    K.igemm(T, GEMM_T2, B, hs, wsz, in0, in1, lv['up_t2'], C, EPI_FINAL,
                            [K.Seg(C, out0=lv['out'], bias=bias, final_act=fa)], ws)
            elif training:
                K.igemm(T, GEMM_T2, B, hs, wsz, in0, in1, lv['up_t2'], C, EPI_Z_STATS,
                        [K.Seg(C, out0=lv['zu'], partials=lv['part_u'])], ws)
                self._bn_forward(lv, 'u', bn, B * 4 * hs * wsz, lv['P_u'], lv['zu'], 0.0, None, lv['ru'])
            else:
                K.bn_eval_affine(bn.weight, bn.bias, bn.running_mean, bn.running_var, BN_EPS, lv['scale_u'],
                                 lv['shift_u'])
                K.igemm(T, GEMM_T2, B, hs, wsz, in0, in1, lv['up_t2'], C, EPI_ACT,
                        [K.Seg(C, out1=lv['ru'], scale=lv['scale_u'], shift=lv['shift_u'])], ws)
        out = L[0]['out']                                   # [B, H, W, Cout] f32
        Cout = L[0]['cu_out']
        if Cout == 1:
            return out.view(B, 1, out.shape[1], out.shape[2])
        res = torch.empty(B, Cout, out.shape[1], out.shape[2], dtype=torch.float32, device=out.device)
        K.nhwc_to_nchw(out, res)
        return res

    # tensors up to this many elements take the one-launch finalize + apply kernels (innermost levels: the two
    # separate launches cost 6-9 us each there, launch-latency bound)
    BN_FUSED_MAX = int(os.environ.get('ADN_BN_FUSED_MAX', 1 << 20))

    def _bn_forward(self, lv, tag, bn, count, P, z, slope, out_leaky, out_relu):
        """Train-mode BatchNorm + activation of a raw conv output: statistics finalize (+ running stats) and apply."""
        track = bn.track_running_stats and bn.running_mean is not None
        C = lv[f'mean_{tag}'].numel()
        args = (lv[f'part_{tag}'], P, C, count, bn.weight, bn.bias, BN_EPS,
                BN_MOMENTUM if bn.momentum is None else bn.momentum,
                bn.running_mean if track else None, bn.running_var if track else None,
                bn.num_batches_tracked if track else None,
                lv[f'mean_{tag}'], lv[f'istd_{tag}'], lv[f'scale_{tag}'], lv[f'shift_{tag}'])
        if count * C <= self.BN_FUSED_MAX and C % 32 == 0:
            K.bn_fwd_fused(*args, z, count, slope, out_leaky, out_relu)
        else:
            K.bn_fwd_finalize(*args)
            K.bn_act(z, count, C, lv[f'scale_{tag}'], lv[f'shift_{tag}'], slope, out_leaky, out_relu)

    def _bn_backward(self, lv, tag, bn, count, P, g, z):
        """BatchNorm backward of the tensor BN'd at (level, tag): dgamma / dbeta + dz in place of g."""
        C = lv[f'mean_{tag}'].numel()
        dg, db = self._flat_slice(self.flat_g, bn.weight), self._flat_slice(self.flat_g, bn.bias)
        if count * C <= self.BN_FUSED_MAX and C % 32 == 0:
            K.bn_bwd_fused(lv[f'bpart_{tag}'], P, C, count, dg, db, g, z, count, lv[f'scale_{tag}'], lv[f'mean_{tag}'],
                           lv[f'istd_{tag}'])
        else:
            K.bn_bwd_finalize(lv[f'bpart_{tag}'], P, C, count, dg, db, lv[f'coef_{tag}'])
            K.bn_bwd_apply(g, z, count, C, lv[f'scale_{tag}'], lv[f'mean_{tag}'], lv[f'istd_{tag}'], lv[f'coef_{tag}'])

    # ------------------------------------------------------------------ backward
    def _ready(self, param):
        if self.on_grad_ready is not None:
            off = self.offset[id(param)]
            _lib.record_py(lambda: self.on_grad_ready(off))

    def dz_target(self):
        """Where a loss kernel may write d loss / d pre-activation of the output directly (adn_loss_finish_dz), together
        with the bias-gradient slot of the last layer: (dz [B, 1, H, W] f32, bias_grad or None, final_act), or None when
        this engine's last layer wants the padded bf16 form (then ``backward`` converts ``gout`` itself)."""
        l0 = self.levels[0]
        if not getattr(self, 'edge_path', False) or l0.get('cu_out') != 1 or os.environ.get('ADN_NO_FUSED_DZ'):
            return None
        bias = l0['up'].bias
        return l0['dz0'], (None if bias is None else self._flat_slice(self.flat_g, bias)), 1 if self.depth_norm else 0

    def backward(self, gout, fused_norm=False, dz_ready=False):
        """gout: d loss / d output, f32 [B, Cout, H, W].  Fills flat_g (all parameters); with ``fused_norm`` also
        ``sq_all`` (see _prepare_fused_norm).  ``dz_ready``: the caller's loss kernel already wrote ``dz_target()``
        (gout is ignored)."""
        T, B, n, L, ws = self.dtype, self.B, self.n, self.levels, self.workspace
        fused_norm = fused_norm and self.sq_all is not None
        l0 = L[0]
        if l0['cu_out'] != 1:
            raise NotImplementedError('backward is implemented for output_nc == 1 (the depth map)')
        up0 = l0['up']
        if not dz_ready:
            gout = gout.contiguous().float()
            K.final_act_bwd(gout, l0['out'], 1 if self.depth_norm else 0, l0['dz0'])
            if up0.bias is not None:
                K.sum_to_scalar(l0['dz0'], self._flat_slice(self.flat_g, up0.bias), self.red_ws)
        # ---- up layers, outermost first
        # (weight gradients of the small-image levels are collected and launched together: nothing inside backward reads
        #  them, and their operands stay untouched until the end of the pass)
        batch = {1: [], 2: []}            # per problem class (kernels.wgrad_batchable)
        held_u, held_d = [], []           # patch-staged groups (see _prepare: pb_u / pb_d)
        # (the reducer's gradient-ready hook wants every dW as early as possible: the same problems then go out as
        #  one-problem launches of the same kernel form -- unsplit --, so both modes produce identical bits)
        use_batch = self.on_grad_ready is None

        def flush(cls=None):
            for c in ((cls,) if cls else (1, 2)):
                if batch[c]:
                    K.wgrad_batch(T, B, batch[c])
                    del batch[c][:]
        for i in range(n):
            lv = L[i]
            hs, wsz = lv['hs'], lv['ws']
            if i == 0:
                dz = lv['dz0']
            else:
                self._bn_backward(lv, 'u', lv['bn_u'], B * 4 * hs * wsz, L[i - 1]['P_gu'], lv['Gu'], lv['zu'])
                dz = lv['Gu']
            in0 = lv['rd']
            in1 = L[i + 1]['ru'] if i < n - 1 else None
            segs = [K.Seg(lv['cd_out'], out0=lv['Gd'], ref=lv['rd'], slope=0.0)]
            if i < n - 1:
                nx = L[i + 1]
                segs.append(K.Seg(nx['cu_out'], out0=nx['Gu'], ref=nx['ru'], slope=0.0, z=nx['zu'],
                                  mean=nx['mean_u'], istd=nx['istd_u'], partials=nx['bpart_u'],
                                  scale=nx['scale_u'], shift=nx['shift_u']))
            if i == 0 and self.edge_path:
                K.thin_wgrad(dz, in0, in1, B, hs, wsz, self._flat_slice(self.flat_g, lv['up'].weight), ws)
                self._ready(lv['up'].weight)
                K.d0_dgrad(dz, self._flat_slice(self.flat_p, lv['up'].weight), B, hs, wsz, segs[0], segs[1])
                continue
            if i in self.pb_u:
                held_u.append((hs, wsz, in0, in1, dz, None, self._flat_slice(self.flat_g, lv['up'].weight),
                               lv['up_sq'] if fused_norm else None))
                if i == self.pb_u[-1]:                    # the group's last (innermost) level: all of them in one launch
                    K.wgrad_patch_batch(T, B, held_u, ws)
                    self._ready(lv['up'].weight)
            elif lv['wb_u']:
                if len(batch[lv['wb_u']]) == 8:
                    flush(lv['wb_u'])
                batch[lv['wb_u']].append((hs, wsz, in0, in1, dz, None, self._flat_slice(self.flat_g, lv['up'].weight),
                                          lv['up_sq'] if fused_norm else None))
                if not use_batch:
                    flush(lv['wb_u'])
            else:
                K.wgrad(T, B, hs, wsz, in0, in1, dz, None, self._flat_slice(self.flat_g, lv['up'].weight), ws,
                        c_valid=lv['cu_out'] if (i == 0 and self.cout_pad != lv['cu_out']) else 0,
                        sq=lv['up_sq'] if fused_norm else None)
            if i not in self.pb_u:
                self._ready(lv['up'].weight)
            K.igemm(T, GEMM_S2, B, hs, wsz, dz, None, lv['up_s2'], lv['cu_in'], EPI_BWD, segs, ws,
                    algo_c=lv['cu_out'])
        # ---- down layers, innermost first
        for i in reversed(range(n)):
            lv = L[i]
            hs, wsz = lv['hs'], lv['ws']
            if lv['bn_d'] is not None:
                self._bn_backward(lv, 'd', lv['bn_d'], B * hs * wsz, L[i + 1]['P_gd'], lv['Gd'], lv['zd'])
            src = self.x_nhwc if i == 0 else L[i - 1]['ad']
            if i == 0 and self.edge_path:
                if self._x_in._version != self._x_ver:
                    raise RuntimeError('the network input was modified in place between forward and backward')
                K.thin_wgrad(self._x_in, lv['Gd'], None, B, hs, wsz, self._flat_slice(self.flat_g, lv['down'].weight), ws)
            elif i in self.pb_d:
                flush()
                held_d.append((hs, wsz, lv['Gd'], None, src, None, self._flat_slice(self.flat_g, lv['down'].weight),
                               lv['down_sq'] if fused_norm else None))
                if i == self.pb_d[0]:                     # the group's last (outermost) level
                    K.wgrad_patch_batch(T, B, held_d, ws)
            elif lv['wb_d']:
                if len(batch[lv['wb_d']]) == 8:
                    flush(lv['wb_d'])
                batch[lv['wb_d']].append((hs, wsz, lv['Gd'], None, src, None,
                                          self._flat_slice(self.flat_g, lv['down'].weight),
                                          lv['down_sq'] if fused_norm else None))
                if not use_batch:
                    flush(lv['wb_d'])
            else:
                flush()                                   # back at the wide levels: the collected small ones go first
                K.wgrad(T, B, hs, wsz, lv['Gd'], None, src, None, self._flat_slice(self.flat_g, lv['down'].weight), ws,
                        c_valid=lv['cd_in'] if (i == 0 and self.cin_pad != lv['cd_in']) else 0,
                        sq=lv['down_sq'] if fused_norm else None)
            if i not in self.pb_d or i == self.pb_d[0]:
                self._ready(lv['down'].weight)
            if i > 0:
                pv = L[i - 1]
                seg = K.Seg(pv['cd_out'], out0=pv['Gd'], ref=pv['ad'], slope=LEAKY, accumulate=True)
                if pv['bn_d'] is not None:
                    seg = K.Seg(pv['cd_out'], out0=pv['Gd'], ref=pv['ad'], slope=LEAKY, accumulate=True,
                                z=pv['zd'], mean=pv['mean_d'], istd=pv['istd_d'], partials=pv['bpart_d'],
                                scale=pv['scale_d'], shift=pv['shift_d'])
                K.igemm(T, GEMM_T2, B, hs, wsz, lv['Gd'], None, lv['down_t2'], lv['cd_in'], EPI_BWD, [seg], ws)
        flush()
        if self.on_grad_ready is not None:
            _lib.record_py(lambda: self.on_grad_ready(0))


class _UNetFunction(torch.autograd.Function):
    """torch.autograd bridge: parameters are inputs so that loss.backward() reaches them."""

    @staticmethod
    def forward(ctx, x, engine, training, *params):
        ctx.engine = engine
        out = engine.forward(x, training)
        return out.clone()          # engine buffers are reused by the next step

    @staticmethod
    def backward(ctx, gout):
        eng = ctx.engine
        eng.backward(gout)
        grads = tuple(eng.grad_view(p) for p, _, _ in eng.param_meta)
        return (None, None, None) + grads


def run_unet(engine, x, training):
    if not engine._bound():
        engine.bind_parameters()
    needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p, _, _ in engine.param_meta)
    if needs_grad and training:
        return _UNetFunction.apply(x, engine, training, *[p for p, _, _ in engine.param_meta])
    with torch.no_grad():
        return engine.forward(x, training).clone()


class GraphedStep:
    """Launch-overhead removal shared by the fused trainers: after ``after_steps`` eager steps the whole step (a few
    hundred libadn launches, nothing synchronising with the host) is captured into ONE hipGraph over static input
    buffers and replayed.  Subclasses implement ``_step_impl(*inputs)`` (inputs may be None) and return device
    tensors that stay valid across replays."""
    _graph = None
    _graph_after = None
    _calls = 0

    def enable_graph(self, after_steps=3):
        self._graph_after = after_steps

    def _graphed(self, *inputs):
        self._calls += 1
        if self._graph is not None:
            if any((b is None) != (t is None) or (b is not None and b.shape != t.shape) for b, t in zip(self._g_in, inputs)):
                return self._step_impl(*inputs)          # another batch shape: eager, on that shape's own buffer set
            for buf, t in zip(self._g_in, inputs):
                if buf is not None:
                    buf.copy_(t)
            self._graph.replay()
            self.engine.weights_dirty, self.engine.s2_fresh = self._post_flags
            return self._g_out
        if self._graph_after is not None and self._calls > self._graph_after:
            self._g_in = [None if t is None else t.contiguous().float().clone() for t in inputs]
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._g_out = self._step_impl(*self._g_in)
            self.engine.pin_buffers()                    # the graph holds raw pointers into this shape's buffer set
            # the replayed optimizer step changes the weights behind Python's back: restore the engine's
            # "operands are stale" flags after every replay (an eval forward in between clears them)
            self._post_flags = (self.engine.weights_dirty, self.engine.s2_fresh)
            self._graph.replay()            # capture only records: run the step once for real
            return self._g_out
        return self._step_impl(*inputs)


class FusedTrainer:
    """One fused training step: forward + masked loss + backward + (all-reduce) + clip + optimizer.

    Mirrors the hot loop of /root/reference/train.py:633-693 (and train_binaural_attention.py:394-433 when
    ``clip_norm`` is None and ``mask_mode`` is 'gt0'; train_rgb_depth.py:355-362 with criterion 'DepthLoss',
    where l1_weight / silog_weight carry lambda_l1 / lambda_smooth).  ``engine`` is a UNetEngine or a DCEngine
    (both expose forward / backward / flat_p / flat_g / flat_w16).  Everything stays on device; ``step`` returns the
    loss as a 0-dim device tensor (call .item() to reproduce the reference's per-step host sync).
    """
    CRIT = {'L1': 0, 'SIlog': 1, 'Combined': 2, 'DepthLoss': 3}
    OPT = {'AdamW': 0, 'Adam': 1, 'SGD': 2}

    def __init__(self, engine, criterion='Combined', l1_weight=0.5, silog_weight=0.5, silog_lambda=0.5,
                 max_depth=30.0, optimizer='AdamW', lr=0.002, betas=(0.9, 0.999), eps=1e-8, weight_decay=None,
                 clip_norm=1.0, mask_mode='ne0', ddp=None):
        self.engine = engine
        self.criterion = self.CRIT[criterion]
        self.l1_weight, self.silog_weight, self.silog_lambda = float(l1_weight), float(silog_weight), float(silog_lambda)
        self.scale = float(max_depth) if engine.depth_norm else 1.0       # train.py:649-652
        self.opt_kind = self.OPT[optimizer]
        self.lr, self.betas, self.eps = float(lr), betas, float(eps)
        if weight_decay is None:                                           # torch defaults (train.py passes only lr)
            weight_decay = 0.01 if optimizer == 'AdamW' else 0.0
        self.weight_decay = float(weight_decay)
        self.clip_norm = clip_norm
        self.mask_mode = 0 if mask_mode == 'ne0' else 1
        self.ddp = ddp
        self._ready = False
        self._graph = None
        self._graph_after = None
        self._plan = None
        self._plan_after = None
        self._calls = 0
        self._gout_valid, self._last_pred_gt = True, None

    def enable_launch_plan(self, after_steps=3):
        """Record the step's launches once (after ``after_steps`` eager steps) and replay the prebuilt ctypes
        calls afterwards: the low-overhead eager mode used with the data-parallel reducer, whose collectives
        stay ordinary torch.distributed calls inside the plan."""
        self._plan_after = after_steps

    def enable_graph(self, after_steps=3):
        """Capture the whole step into one hipGraph after ``after_steps`` eager steps (fixed batch shape).

        Every launch of the step is on the current stream and nothing synchronises with the host, so the
        ~140 kernel launches replay as one graph launch.  Not combined with the data-parallel reducer.
        """
        if self.ddp is not None:
            raise RuntimeError('graph capture of the step is only wired for single-process training')
        self._graph_after = after_steps

    def _setup(self, dev):
        eng = self.engine
        if not eng._bound():
            eng.bind_parameters()
        f64 = dict(dtype=torch.float64, device=dev)
        self.stats = torch.zeros(4, **f64)
        self.state = torch.zeros(8, **f64)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.loss_ws = torch.empty(4096 + 8, **f64)
        self.norm_ws = torch.empty(1024 + 8, **f64)
        self.exp_avg = torch.zeros_like(eng.flat_p)
        self.exp_avg_sq = torch.zeros_like(eng.flat_p)
        self.gout = None
        self._gout_sets = {}
        self._flat_id = eng.flat_p.data_ptr()
        self.bucket_norm = None
        if self.ddp is not None:
            self.ddp.attach(eng)
            if self.clip_norm is not None and eng.flat_g.is_cuda:
                self.bucket_norm = self.ddp.enable_bucket_norm()      # sums of squares per reduced bucket (ddp.finish)
        self._ready = True

    def state_dict(self):
        """The checkpoint's 'optimizer' / 'optimizer_state_dict' entry in ``torch.optim`` format (optim_state.py): what
        ``torch.optim.AdamW(model.parameters(), ...).state_dict()`` would hold after the same steps, so the reference's
        ``optimizer.load_state_dict`` (train_binaural_attention.py:361) reads it and vice versa."""
        from . import optim_state
        eng = self.engine
        if not eng._bound():
            eng.bind_parameters()
        step = int(self.state[0].item()) if self._ready else 0
        return optim_state.export_state(eng.param_meta, eng._view, self.exp_avg if self._ready else None,
                                        self.exp_avg_sq if self._ready else None, step, self.opt_kind, self.lr,
                                        self.betas, self.eps, self.weight_decay)

    def load_state_dict(self, sd, device):
        """Restore a ``torch.optim`` state dict (written by this class or by the reference's torch optimizer over the
        same parameters); the flat layout of round 1 ('exp_avg' / 'exp_avg_sq' / 'step')
        is re-sliced parameter by parameter (its alignment was 4 elements, today's is 8) or rejected."""
        from . import optim_state
        self._setup(device)
        if optim_state.is_torch_format(sd):
            step, group = optim_state.import_state(sd, self.engine.param_meta, self.engine._view, self.exp_avg,
                                                   self.exp_avg_sq)
            optim_state.adopt_group(self, group)
            self._set_step(step)
        elif 'exp_avg' in sd:
            step = optim_state.import_legacy_flat(sd, self.engine.param_meta, self.exp_avg, self.exp_avg_sq)
            self._set_step(step)

    def _set_step(self, step):
        """state = [step, 1 - beta1^step, 1 - beta2^step, ...] (adn_optimizer_step advances all three)."""
        self.state[0] = float(step)
        self.state[1] = 1.0 - self.betas[0] ** step
        self.state[2] = 1.0 - self.betas[1] ** step

    def step(self, audio, gt):
        self._calls += 1
        captured = self._plan is not None or self._graph is not None
        if captured and (audio.shape != self._g_audio.shape or gt.shape != self._g_gt.shape):
            return self._step_impl(audio, gt)            # another batch shape: eager, on that shape's own buffer set
        if self._plan is not None:
            self._g_audio.copy_(audio)
            self._g_gt.copy_(gt)
            _lib.replay(self._plan)
            self.engine.weights_dirty, self.engine.s2_fresh = self._post_flags
            return self._g_out
        if self._plan_after is not None and self._calls > self._plan_after and self._ready:
            self._g_audio, self._g_gt = audio.clone(), gt.contiguous().float().clone()
            _lib.RECORD = []
            try:
                self._g_out = self._step_impl(self._g_audio, self._g_gt)
                self._plan = _lib.RECORD
            finally:
                _lib.RECORD = None
            self.engine.pin_buffers()                    # the plan holds raw pointers into this shape's buffer set
            self._post_flags = (self.engine.weights_dirty, self.engine.s2_fresh)
            return self._g_out
        if self._graph is not None:
            self._g_audio.copy_(audio)
            self._g_gt.copy_(gt)
            self._graph.replay()
            self.engine.weights_dirty, self.engine.s2_fresh = self._post_flags
            return self._g_out
        if self._graph_after is not None and self._calls > self._graph_after and self._ready:
            self._g_audio, self._g_gt = audio.clone(), gt.contiguous().float().clone()
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._g_out = self._step_impl(self._g_audio, self._g_gt)
            self.engine.pin_buffers()                    # the graph holds raw pointers into this shape's buffer set
            self._post_flags = (self.engine.weights_dirty, self.engine.s2_fresh)
            self._graph.replay()            # capture only records: run the step once for real
            return self._g_out
        return self._step_impl(audio, gt)

    def loss_gradient(self):
        """d loss / d prediction of the last step (diagnostics, tests).  When the fused loss kernel wrote the gradient of
        the output's PRE-activation instead (adn_loss_finish_dz), it is recomputed here from that step's statistics."""
        if not self._gout_valid:
            pred, gt = self._last_pred_gt
            K.loss_finish(pred, gt, self.scale, self.mask_mode, 1e-6, self.stats, self.criterion, self.l1_weight,
                          self.silog_weight, self.silog_lambda, None, self.gout)
            self._gout_valid = True
        return self.gout

    def _step_impl(self, audio, gt):
        eng = self.engine
        if not self._ready or self._flat_id != (eng.flat_p.data_ptr() if eng.flat_p is not None else None):
            self._setup(audio.device)
        pred = eng.forward(audio, True)
        gt = gt.contiguous().float()
        dz_ready = False
        # loss-gradient scratch PER BATCH SHAPE, never freed: a captured hipGraph / launch plan holds its raw pointer, and an
        # eager step on another batch shape in between (ragged last batch) must not hand that block back to the allocator
        key = tuple(pred.shape)
        if key not in self._gout_sets:
            self._gout_sets[key] = torch.empty_like(pred)
        self.gout = self._gout_sets[key]
        if self.criterion == 3:       # DepthLoss: unmasked L1 + total variation (train_rgb_depth.py:43-87)
            K.l1tv_stats(pred, gt, self.stats, self.loss_ws)
            if self.ddp is not None:
                _lib.record_py(lambda: self.ddp.all_reduce_loss_stats(self.stats))
            K.l1tv_finish(pred, gt, self.stats, self.ddp.world_size if self.ddp is not None else 1, self.l1_weight,
                          self.silog_weight, self.loss, self.gout)
        else:
            K.loss_stats(pred, gt, self.scale, self.mask_mode, 1e-6, self.stats, self.loss_ws)
            if self.ddp is not None:      # one global-batch loss, as under DataParallel
                _lib.record_py(lambda: self.ddp.all_reduce_loss_stats(self.stats))
            # U-Net with the thin last layer: the loss kernel writes d loss / d pre-activation and the bias gradient itself
            target = eng.dz_target() if (hasattr(eng, 'dz_target') and self.criterion <= 2) else None
            if target is not None:
                dz, bias_grad, final_act = target
                K.loss_finish_dz(pred, gt, self.scale, self.mask_mode, 1e-6, self.stats, self.criterion, self.l1_weight,
                                 self.silog_weight, self.silog_lambda, self.loss, dz, final_act, bias_grad, self.loss_ws)
                dz_ready = True
            else:
                K.loss_finish(pred, gt, self.scale, self.mask_mode, 1e-6, self.stats, self.criterion, self.l1_weight,
                              self.silog_weight, self.silog_lambda, self.loss, self.gout)
        if self.ddp is not None:
            _lib.record_py(self.ddp.begin_backward)
        # single process: the weight-gradient kernels leave their share of the total norm behind (no pass over flat_g);
        # under the reducer the norm is that of the all-reduced gradients, taken afterwards
        fused = (self.clip_norm is not None and self.ddp is None and getattr(eng, 'supports_fused_norm', False)
                 and eng.sq_all is not None)
        self._gout_valid, self._last_pred_gt = not dz_ready, (pred, gt)
        if dz_ready:
            eng.backward(self.gout, fused_norm=fused, dz_ready=True)
        elif fused:
            eng.backward(self.gout, fused_norm=True)
        else:
            eng.backward(self.gout)
        if self.ddp is not None:
            _lib.record_py(self.ddp.finish)
        if fused:
            K.grad_norm_ranges(eng.flat_g, eng.norm_ranges, eng.sq_all, float(self.clip_norm), self.state, self.norm_ws)
        elif self.clip_norm is not None and self.bucket_norm is not None:
            K.grad_norm_ranges(eng.flat_g, None, self.bucket_norm, float(self.clip_norm), self.state, self.norm_ws)
        elif self.clip_norm is not None:
            K.grad_norm(eng.flat_g, float(self.clip_norm), self.state, self.norm_ws)
        K.optimizer_step(eng.flat_p, eng.flat_g, self.exp_avg, self.exp_avg_sq, self.opt_kind, self.lr,
                         self.betas[0], self.betas[1], self.eps, self.weight_decay, self.clip_norm is not None,
                         self.state, bf16_copy=eng.flat_w16)
        eng.weights_dirty = True
        eng.s2_fresh = eng.flat_w16 is not None      # the optimizer just refreshed the bf16 S2 operands
        return self.loss[0], pred
