"""oracle/mx8_oracle.py (the MX-fp8 restatement behind tests/test_gpu_mx8.py) pinned on the CPU: element rounding against
torch's float8_e4m3fn conversion, the E8M0 rule and the pack layout against known answers."""
import numpy as np
import torch

from oracle import mx8_oracle as mx


def test_e4m3_rounding_matches_torch_float8():
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.standard_normal(20000) * np.exp2(rng.integers(-12, 9, 20000)),
                        np.array([0.0, -0.0, 448.0, -448.0, 2.0 ** -9, 2.0 ** -10, 3 * 2.0 ** -10, 2.0 ** -6, 15.5, 17.0, 19.0]),
                        np.arange(-4480, 4481) / 10.0])
    v = np.clip(v, -448, 448).astype(np.float32)
    want = torch.from_numpy(v).to(torch.float8_e4m3fn)
    q = mx.e4m3_round(v)
    np.testing.assert_array_equal(q.astype(np.float32), want.float().numpy())
    bits = mx.e4m3_bits(q)
    wb = want.view(torch.uint8).numpy()
    nz = q != 0                                           # (the sign of a zero is not part of the contract)
    np.testing.assert_array_equal(bits[nz], wb[nz])
    np.testing.assert_array_equal(mx.e4m3_decode(bits), q)
    assert float(mx.e4m3_round(1e6)) == 448.0 and float(mx.e4m3_round(-500.0)) == -448.0      # saturating


def test_e8m0_rule_known_answers():
    # amax 1.0 -> 2^(0 - 8): byte 119;  448 = 1.75 * 2^8 -> byte 127;  449 -> 128 (no saturation);  300 -> 127;  512 -> 128
    np.testing.assert_array_equal(mx.e8m0_byte([1.0, 448.0, 449.0, 300.0, 512.0, 0.0, 2.0 ** -130, 3e38]),
                                  [119, 127, 128, 127, 128, 0, 0, 247])
    x = np.zeros((1, 64), np.float32)
    x[0, :32] = np.linspace(-300, 300, 32)
    x[0, 32] = 2.0 ** -20
    bits, byte, deq = mx.quantize(x)
    assert byte.tolist() == [[127, 99]]                   # second block: amax 2^-20 -> scale 2^-28, element = 2^8 = 256
    assert float(deq[0, 32]) == 2.0 ** -20 and float(mx.e4m3_decode(bits[0, 32])) == 256.0
    # relative error of a block's elements: within half an e4m3 ulp of the scaled value (2^-4 relative for normals)
    big = np.abs(x[0, :32]) >= 300 * 2.0 ** -6
    assert np.all(np.abs(deq[0, :32] - x[0, :32])[big] <= 2.0 ** -4 * np.abs(x[0, :32])[big])


def test_pack_layout_and_dummy_tap():
    rng = np.random.default_rng(1)
    w = rng.standard_normal((64, 128, 3, 3)).astype(np.float32)
    w8, wsc, deq = mx.pack_weights(w, transpose=False)
    assert w8.shape == (64, 10, 128) and wsc.shape == (64, 2, 5, 4) and deq.shape == (64, 9, 128)
    assert not w8[:, 9].any() and (wsc[:, :, 4, 2:] == 127).all()
    # tap 5 (ky 1, kx 2), channels 32..63 of chunk 1: pair 2, slot (5 & 1) * 2 + 1 = 3
    blk = w[:, 96:128, 1, 2]
    np.testing.assert_array_equal(wsc[:, 1, 2, 3], mx.e8m0_byte(np.abs(blk).max(axis=1)))
    np.testing.assert_allclose(deq[:, 5, 96:128], blk, rtol=2.0 ** -4, atol=1e-3)
    w8t, wsct, deqt = mx.pack_weights(w, transpose=True)
    assert w8t.shape == (128, 10, 64) and wsct.shape == (128, 1, 5, 4)
    np.testing.assert_allclose(deqt[:, 8 - 5, :], w[:, :, 1, 2].T, rtol=2.0 ** -4, atol=1e-3)     # flipped taps
