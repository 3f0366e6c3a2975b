"""Brute-force check of Mth<float, true>::cell_coord (rollout_fwd_kernel.h): u = a * y, u2 = fma(fma(-u, b, a), y, u) with y = RN(1 / b)
equals the IEEE float32 quotient a / b -- exhaustively over every float32 a in [1e-3, 13) for the two map resolutions of the BASELINE
configs, and over every 7th float32 in [1e-3, 64) for other resolutions.  Pure numpy (float64 emulates the fused multiply-adds
exactly: products of two float32 have 48 significant bits).  Expected output: zero mismatches everywhere."""
import numpy as np, sys
def check(res, lo_bits, hi_bits, step=1):
    b = np.float32(res)
    y = np.float32(1.0 / np.float64(b))
    bad = 0; worst=None
    CH = 1 << 24
    for s in range(lo_bits, hi_bits, CH*step):
        bits = np.arange(s, min(s + CH*step, hi_bits), step, dtype=np.int64).astype(np.uint32)
        a = bits.view(np.float32)
        true = (a / b)            # IEEE f32 division (numpy float32 op)
        q = (a * y)               # f32 mul
        r = (a.astype(np.float64) - q.astype(np.float64) * np.float64(b))   # exact
        r32 = r.astype(np.float32)
        assert np.all(r32.astype(np.float64) == r)
        q2 = (q.astype(np.float64) + r * np.float64(y)).astype(np.float32)
        m = q2 != true
        bad += int(m.sum())
        if m.any() and worst is None: worst = (a[m][0], true[m][0], q2[m][0], q[m][0])
        # also count how often the plain product differs
    return bad, worst
for res in [0.05, 0.1]:
    lo = np.float32(1e-3).view(np.uint32); hi = np.float32(13.0).view(np.uint32)
    print(res, 'exhaustive [1e-3,13):', check(res, int(lo), int(hi)))
rng = np.random.default_rng(0)
for res in [0.2, 0.025, 0.07, 0.3, 0.0333, 0.125, 0.4]:
    lo = np.float32(1e-3).view(np.uint32); hi = np.float32(64.0).view(np.uint32)
    print(res, 'strided:', check(res, int(lo), int(hi), step=7))
