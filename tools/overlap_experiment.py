"""Exchange beside the backward pass, or after it?  A one-GPU experiment.

    python tools/overlap_experiment.py

The data-parallel design (ddp.py) overlaps the 217.6 MB gradient all-reduce with the backward pass.  On one GPU the local
side of that exchange is emulated by `adn_debug_stream_rmw`: K workgroups (what an RCCL ring kernel occupies: 16 / 32 / 64)
that stream the gradient buffer through HBM (read + read-modify-write = 3 x 217.6 MB of traffic per exchange) on a SIDE
stream, started when the backward pass starts.  Reported per K: the step time with the stream beside the backward, the
stream's own duration alone, and the two policies' totals
    overlapped  = step time with the side stream running (the step ends when both are done)
    serial      = plain step + stream alone
"""
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402

import torch  # noqa: E402

from audio_depth_estimation_amd import _lib  # noqa: E402
from audio_depth_estimation_amd.engine import FusedTrainer  # noqa: E402
from audio_depth_estimation_amd.models.unetbaseline_model import define_G  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = define_G(SimpleNamespace(dataset=SimpleNamespace(depth_norm=False, max_depth=30.0)), 2, 1, 64, 'unet_256')
    model.compute_dtype = torch.bfloat16
    model = model.to(dev).train()
    eng = model.engine()
    tr = FusedTrainer(eng, 'Combined', 0.237, 0.637, 0.869, max_depth=30.0, optimizer='AdamW', lr=0.002, clip_norm=1.0)
    g = torch.Generator().manual_seed(1234)
    audio = torch.rand(32, 2, 256, 256, generator=g).to(dev)
    gt = 30.0 * torch.rand(32, 1, 256, 256, generator=g)
    gt[gt < 3.0] = 0.0
    gt = gt.to(dev)
    for _ in range(3):
        tr.step(audio, gt)
    nbytes = eng.flat_g.numel() * 4 // 16 * 16
    src = torch.empty(nbytes // 4, device=dev)
    dst = torch.zeros(nbytes // 4, device=dev)
    side = torch.cuda.Stream()
    lib = _lib.load()
    state = {'k': 0}
    orig_backward = eng.backward

    def backward(*a, **kw):
        if state['k']:
            ev = torch.cuda.Event()
            ev.record()                               # the exchange may start when the backward pass starts
            side.wait_event(ev)
            _lib.check(lib.adn_debug_stream_rmw(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), nbytes, state['k'], 1,
                                                C.c_void_p(side.cuda_stream)), 'adn_debug_stream_rmw')
        return orig_backward(*a, **kw)

    eng.backward = backward

    def time_steps(n=30):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            tr.step(audio, gt)
            torch.cuda.current_stream().wait_stream(side)      # the optimizer of the NEXT step needs the exchanged gradients
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    def time_stream(k, n=10):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            _lib.check(lib.adn_debug_stream_rmw(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), nbytes, k, 1,
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'adn_debug_stream_rmw')
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    state['k'] = 0
    plain = time_steps()
    print(f'plain eager step (no exchange): {plain:.3f} ms   [gradient buffer {nbytes / 1e6:.1f} MB, stream traffic {3 * nbytes / 1e6:.0f} MB]')
    for k in (16, 32, 64, 128):
        alone = time_stream(k)
        state['k'] = k
        both = time_steps()
        state['k'] = 0
        print(f'K = {k:3d} workgroups: stream alone {alone:.3f} ms ({3 * nbytes / alone / 1e9:.2f} TB/s)   step with the stream beside the backward '
              f'{both:.3f} ms (+{both - plain:.3f})   serial policy {plain + alone:.3f} ms   -> overlap {"wins" if both < plain + alone else "loses"} by '
              f'{abs(plain + alone - both):.3f} ms', flush=True)


if __name__ == '__main__':
    main()
