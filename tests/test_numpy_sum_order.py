"""The order of numpy's float32 sums, which the device epilogues restate (np_sum_f32, np_sum_f32_wave,
np_sum8 in gps-sdr-receiver_amd/csrc/gpsmi_trk.hip): for 8 <= n < 128 elements numpy adds eight strided
accumulators r[j] = a[j] + a[j + 8] + ..., combines them as ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7))
and adds the n % 8 tail elements in order; below eight elements it adds in order.  The reference takes
np.mean / np.std of 32 or 33 float32 values per block and channel (gpslib.py:1186-1187, :1215-1262); the
kernels reproduce those bits only because this recipe is numpy's."""
import numpy as np
import pytest


def device_sum(a):
    """np_sum_f32 of the kernels, in float32 scalar arithmetic."""
    a = np.asarray(a, dtype=np.float32)
    n = len(a)
    if n < 8:
        r = np.float32(0)
        for v in a:
            r = np.float32(r + v)
        return r
    r = [a[j] for j in range(8)]
    body = n - n % 8
    for i in range(8, body, 8):
        for j in range(8):
            r[j] = np.float32(r[j] + a[i + j])
    res = np.float32(np.float32(np.float32(r[0] + r[1]) + np.float32(r[2] + r[3])) +
                     np.float32(np.float32(r[4] + r[5]) + np.float32(r[6] + r[7])))
    for i in range(body, n):
        res = np.float32(res + a[i])
    return res


def lanes_sum(a):
    """np_sum8 of trk_epilogue8_kernel: element i in slot i // 8 of lane i % 8, n = 8 K' or 8 K' + 1."""
    a = np.asarray(a, dtype=np.float32)
    n = len(a)
    slots = n // 8
    lane = [a[j] for j in range(8)]
    for k in range(1, slots):
        for j in range(8):
            lane[j] = np.float32(lane[j] + a[8 * k + j])
    for step in (1, 2):                       # quad_perm [1,0,3,2], quad_perm [2,3,0,1]
        lane = [np.float32(lane[j] + lane[j ^ step]) for j in range(8)]
    lane = [np.float32(lane[j] + lane[7 - j]) for j in range(8)]      # row_half_mirror
    assert len({v.tobytes() for v in lane}) == 1                        # every lane of the group holds it
    res = lane[0]
    if n % 8:
        res = np.float32(res + a[8 * slots])
    return res


@pytest.mark.parametrize('n', [1, 4, 7, 8, 9, 16, 17, 32, 33, 40, 64, 127])
def test_float32_sum_order_is_numpys(n):
    rng = np.random.default_rng(1000 + n)
    for trial in range(50):
        scale = np.float32(10.0 ** rng.integers(-3, 4))
        a = (rng.standard_normal(n).astype(np.float32) * scale +
             (np.float32(trial % 3) * scale)).astype(np.float32)
        want = np.sum(a)
        assert want.dtype == np.float32
        assert device_sum(a).tobytes() == want.tobytes(), (n, trial)
        assert np.float32(device_sum(a) / np.float32(n)).tobytes() == np.mean(a).tobytes(), (n, trial)
        if n >= 8 and n % 8 <= 1:
            assert lanes_sum(a).tobytes() == want.tobytes(), (n, trial)


def test_std_is_the_mean_of_squared_deviations():
    """STD_DEV = np.std(|g|) as the epilogue forms it: sqrt(sum((x - mean)^2) / n), each step float32."""
    rng = np.random.default_rng(7)
    for n in (32, 33):
        for _ in range(50):
            x = np.abs(rng.standard_normal(n).astype(np.float32) * np.float32(0.05))
            m = np.float32(device_sum(x) / np.float32(n))
            dev = (x - m).astype(np.float32)
            sq = (dev * dev).astype(np.float32)
            got = np.sqrt(np.float32(device_sum(sq) / np.float32(n)))
            assert np.float32(got).tobytes() == np.std(x).tobytes()
