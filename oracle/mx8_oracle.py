"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the block-scaled fp8 format the MX path of libadn
computes in (csrc/mx8.hip) -- never imported by the product path.

The reference has no fp8 code (BASELINE.json configs[4] names "fp8 MFMA conv path" as the MI355X target precision of
models/rgb_depth_model.py:148-218); the format restated here is the published OCP Microscaling (MX) v1.0 MXFP8:
  * elements: OCP e4m3fn (bias 7, 3 mantissa bits, max 448, no infinities), round to nearest even, saturating;
  * one E8M0 scale 2^(byte - 127) per block of 32 consecutive channels: the smallest power of two >= amax / 448, so
    that no element saturates (amax = m 2^e: exponent e - 8 for m <= 1.75, else e - 7; the floor rule printed in the MX
    v1.0 text, e - 8 always, clips elements with m > 1.75 by up to 12.5 %), clamped to [0, 254]; 0 for an all-zero block.
Pinned by: torch's float8_e4m3fn conversion for the element rounding (tests/test_mx8_oracle.py) and known-answer vectors.
The convolution oracle dequantises to float64 and runs the float64 convolution: the HIP kernel's products are exact and only
its f32 accumulation order and the bf16 rounding of the stored output differ.
"""
import numpy as np

E4M3_MAX = 448.0


def e8m0_byte(amax):
    """E8M0 scale byte of blocks with absolute maxima ``amax`` (float32 semantics of the exponent extraction)."""
    amax = np.asarray(amax, dtype=np.float32)
    bits = amax.view(np.uint32)
    e = ((bits >> 23) & 0xff).astype(np.int64) - 8 + ((bits & 0x7fffff) > 0x600000)
    byte = np.clip(e, 0, 254)
    return np.where(amax > 0, byte, 0).astype(np.uint8)


def e4m3_round(v):
    """Round float64 values to the e4m3fn grid (RNE, saturating at +-448); returns float64 values on the grid."""
    v = np.clip(np.asarray(v, dtype=np.float64), -E4M3_MAX, E4M3_MAX)
    a = np.abs(v)
    with np.errstate(divide='ignore'):
        e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    e = np.maximum(e, -6.0)                              # subnormals share the exponent of the smallest normal
    step = np.exp2(e - 3.0)
    q = np.rint(a / step) * step                         # np.rint = round half to even
    return np.sign(v) * q


def e4m3_bits(q):
    """Encode values that lie on the e4m3fn grid."""
    q = np.asarray(q, dtype=np.float64)
    a = np.abs(q)
    sign = (np.signbit(q)).astype(np.uint8) << 7
    with np.errstate(divide='ignore'):
        e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    normal = a >= 2.0 ** -6
    eb = np.where(normal, e + 7, 0).astype(np.int64)
    man = np.where(normal, np.rint((a / np.exp2(e) - 1.0) * 8.0), np.rint(a / 2.0 ** -9)).astype(np.int64)
    return (sign | (eb << 3).astype(np.uint8) | man.astype(np.uint8)).astype(np.uint8)


def e4m3_decode(bits):
    bits = np.asarray(bits, dtype=np.uint8).astype(np.int64)
    s = np.where(bits & 0x80, -1.0, 1.0)
    eb, man = (bits >> 3) & 0xf, bits & 7
    return s * np.where(eb > 0, np.exp2(eb - 7.0) * (1.0 + man / 8.0), man * 2.0 ** -9)


def quantize(x):
    """x [..., C] (C % 32 == 0; values as float32, e.g. upcast bf16) -> (bits uint8 [..., C], scales uint8 [..., C/32],
    dequantised float64 [..., C])."""
    x = np.asarray(x, dtype=np.float32)
    shp = x.shape
    blk = x.reshape(-1, shp[-1] // 32, 32).astype(np.float64)
    byte = e8m0_byte(np.abs(blk).max(axis=-1).astype(np.float32))
    scale = np.exp2(byte.astype(np.float64) - 127.0)[..., None]
    q = e4m3_round(blk / scale)
    return e4m3_bits(q).reshape(shp), byte.reshape(shp[:-1] + (shp[-1] // 32,)), (q * scale).reshape(shp)


def pack_weights(w, transpose):
    """w [X, Y, 3, 3] float32 -> (w8 [rows][10][K], wsc [rows][K/64][5][4], dequantised float64 [rows][9][K]) with
    rows, K = (X, Y) forward / (Y, X) input gradient (taps flipped), as adn_mx8_pack."""
    w = np.asarray(w, dtype=np.float32)
    X, Y = w.shape[:2]
    m = w.transpose(0, 2, 3, 1).reshape(X, 9, Y)                 # master [X][tap][Y]
    if transpose:
        m = m[:, ::-1, :].transpose(2, 1, 0)                     # [Y][8 - tap][X]
    rows, _, K = m.shape
    bits, byte, deq = quantize(np.ascontiguousarray(m))
    w8 = np.zeros((rows, 10, K), np.uint8)
    w8[:, :9] = bits
    wsc = np.full((rows, K // 64, 5, 4), 127, np.uint8)
    for tap in range(9):
        for blk in range(K // 32):
            wsc[:, blk >> 1, tap >> 1, (tap & 1) * 2 + (blk & 1)] = byte[:, tap, blk]
    return w8, wsc, deq


def conv3x3(x_deq, w_deq_fwd):
    """x_deq [B,H,W,C] float64 (dequantised activations), w_deq_fwd [N][9][C] float64 -> [B,H,W,N] float64."""
    import torch
    x = torch.from_numpy(np.ascontiguousarray(x_deq)).permute(0, 3, 1, 2)
    N, _, Cc = w_deq_fwd.shape
    w = torch.from_numpy(np.ascontiguousarray(w_deq_fwd)).reshape(N, 3, 3, Cc).permute(0, 3, 1, 2)
    return torch.nn.functional.conv2d(x, w, padding=1).permute(0, 2, 3, 1).contiguous().numpy()
