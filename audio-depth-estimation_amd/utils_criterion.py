"""Evaluation metrics (mirror of the reference's utils_criterion.py).

``compute_errors(gt, pred, min_depth_threshold=0.0) -> (abs_rel, rmse, a1, a2, a3, log_10, mae)`` keeps the
reference signature and every branch of /root/reference/utils_criterion.py:6-90, but the masked reductions
run in one libadn kernel on the device (no per-sample .cpu().numpy() round trip).  Accepts numpy arrays or
torch tensors; host inputs are uploaded -- there is no numpy implementation in the product path.
``compute_errors_batch`` is the batched validation form: one 7-vector per sample in a single launch.
"""
import numpy as np
import torch

from . import kernels as K


def _device_f32(x, dev):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x.detach().to(device=dev, dtype=torch.float32).contiguous()


def _pick_device(*xs):
    for x in xs:
        if torch.is_tensor(x) and x.is_cuda:
            return x.device
    if not torch.cuda.is_available():
        raise RuntimeError('compute_errors runs on the libadn HIP kernel: no HIP device is visible')
    return torch.device('cuda', torch.cuda.current_device())


def compute_errors_batch(gt, pred):
    """gt, pred: [B, ...] -> float32 tensor [B, 7] on the device (per-sample metrics)."""
    dev = _pick_device(gt, pred)
    g, p = _device_f32(gt, dev), _device_f32(pred, dev)
    B = g.shape[0]
    out = torch.empty(B, 7, dtype=torch.float32, device=dev)
    K.compute_errors(g.view(B, -1), p.reshape(B, -1), out)
    return out


def compute_errors(gt, pred, min_depth_threshold=0.0):
    """All elements of gt/pred form ONE set (as the reference: boolean-mask flattening), 7 floats back."""
    dev = _pick_device(gt, pred)
    g, p = _device_f32(gt, dev), _device_f32(pred, dev)
    out = torch.empty(1, 7, dtype=torch.float32, device=dev)
    K.compute_errors(g.view(1, -1), p.reshape(1, -1), out)
    vals = out[0].tolist()
    return tuple(float(v) for v in vals)
