"""Philox4x32-10 in numpy (Salmon et al., SC'11; Random123 constants) and the product's bits -> uniform convention
(include/soccer_hip.h, ABI 3), written from the specification and NOT through oracle/: the tests that pin the HIP kernels
straight to the reference's fixtures use this, so that nothing of the CPU restatement sits between the kernel and the
reference-held data.  tests/test_host_logic.py checks it against the Random123 known-answer vectors."""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Counter words as arrays (broadcast together), key words as Python ints.  Returns uint32[4, ...]."""
    c0, c1, c2, c3 = np.broadcast_arrays(*(np.asarray(c, dtype=np.uint64) & _MASK for c in (c0, c1, c2, c3)))
    c0, c1, c2, c3 = c0.copy(), c1.copy(), c2.copy(), c3.copy()
    k0 &= 0xFFFFFFFF; k1 &= 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0; p1 = _M1 * c2
        n0 = (p1 >> _S32) ^ c1 ^ np.uint64(k0)
        n2 = (p0 >> _S32) ^ c3 ^ np.uint64(k1)
        c1 = p1 & _MASK; c3 = p0 & _MASK; c0 = n0; c2 = n2
        k0 = (k0 + _W0) & 0xFFFFFFFF; k1 = (k1 + _W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3]).astype(np.uint32)


def lane_words(seed, lanes, c, purpose=0):
    """The word global lane g owns in block(c, purpose): four consecutive lanes share a block, lane g takes word g & 3."""
    g = np.asarray(lanes, dtype=np.uint64)
    q = g >> np.uint64(2)
    c = np.asarray(c, dtype=np.uint64)
    blk = philox4x32_10(q & _MASK, q >> _S32, c & _MASK, (c >> _S32) | np.uint64(purpose << 31),
                        int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    return blk[(g & np.uint64(3)).astype(np.intp), np.arange(g.size)]


def step_draws(seed, lanes, tick, slip_prob):
    """(u_step, u_reset) float64 of every lane at `tick`: u = (m + 1/2) * 2^-b.
    slip_prob > 0: block(tick), m = w >> 2 (b = 30) / w & 3 (b = 2);
    slip_prob == 0: block(tick >> 3), nibble (tick & 7) ^ 1 of w: m = nib >> 2 / nib & 3 (b = 2)."""
    if slip_prob != 0.0:
        w = lane_words(seed, lanes, tick)
        return ((w >> 2).astype(np.float64) + 0.5) * 2.0 ** -30, ((w & 3).astype(np.float64) + 0.5) * 0.25
    w = lane_words(seed, lanes, tick >> 3)
    nib = (w >> np.uint32(4 * ((tick & 7) ^ 1))) & np.uint32(15)
    return ((nib >> 2).astype(np.float64) + 0.5) * 0.25, ((nib & 3).astype(np.float64) + 0.5) * 0.25
