"""The hardware rule the k-means filter's certificate rests on (pmarlo_amd/csrc/kmeans_filter.h, step 1):
v_mfma_f32_16x16x32_bf16 sums its 32 products and C with an error of at most 33 x 2^-24 of the largest term,
in any order of the slots.  The bound was measured on one MI355X (tools/probe/bf16_filter_probe.hip); this
test repeats the measurement through the C ABI on every box the suite runs on, so that a part with a narrower
internal accumulator fails here and not in a label."""

from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

U = 2.0 ** -24


def _bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit patterns, round to nearest even (finite input)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = u + 0x7FFF + ((u >> 16) & 1)
    return (u >> 16).astype(np.uint16)


def _bf16_val(bits: np.ndarray) -> np.ndarray:
    return (bits.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def _exact(a_bits, b_bits, c):
    """Products of two bf16 numbers are exact in float64; math.fsum gives the correctly rounded sum."""
    import math

    a, b = _bf16_val(a_bits), _bf16_val(b_bits)
    t = a.shape[0]
    out = np.empty((t, 16, 16))
    big = np.empty((t, 16, 16))
    for s in range(t):
        for i in range(16):
            for j in range(16):
                terms = list(a[s, i, :] * b[s, :, j]) + [float(c[s, i, j])]
                out[s, i, j] = math.fsum(terms)
                big[s, i, j] = max(abs(v) for v in terms)
    return out, big


def test_accumulation_error_bound(engine):
    """Random operands over a wide range of magnitudes and signs: |D - exact| <= 33 x 2^-24 x the largest term
    (the filter assumes 34.3 per instruction: the bound plus the final rounding)."""
    rng = np.random.default_rng(2024)
    t = 24
    worst = 0.0
    for spread in (0.0, 4.0, 12.0, 30.0):
        a = rng.normal(size=(t, 16, 32)) * np.exp2(rng.uniform(-spread, spread, size=(t, 16, 32)))
        b = rng.normal(size=(t, 32, 16)) * np.exp2(rng.uniform(-spread, spread, size=(t, 32, 16)))
        c = (rng.normal(size=(t, 16, 16)) * np.exp2(rng.uniform(-spread, spread, size=(t, 16, 16)))).astype(np.float32)
        ab, bb = _bf16_bits(a), _bf16_bits(b)
        got = engine.mfma_bf16_probe(ab, bb, c).astype(np.float64)
        want, big = _exact(ab, bb, c)
        err = np.abs(got - want) / (big * U)
        worst = max(worst, float(err.max()))
    assert worst <= 33.0, f"accumulation error {worst:.2f} x 2^-24 of the largest term exceeds the filter's bound"


def test_any_slot_order_stays_inside_the_bound(engine):
    """The filter orders its 64 slots for the layout of the frame images (kmeans_filter.h, filter_slot).  The sum is
    NOT bit-identical under a permutation of the slots (the instruction adds groups of products, each group rounded
    on its own), so what the certificate may use is the bound, for every order."""
    rng = np.random.default_rng(7)
    t = 8
    a = _bf16_bits(rng.normal(size=(t, 16, 32)) * np.exp2(rng.uniform(-10, 10, size=(t, 16, 32))))
    b = _bf16_bits(rng.normal(size=(t, 32, 16)) * np.exp2(rng.uniform(-10, 10, size=(t, 32, 16))))
    c = rng.normal(size=(t, 16, 16)).astype(np.float32)
    want, big = _exact(a, b, c)
    worst = 0.0
    for trial in range(5):
        perm = np.arange(32) if trial == 0 else rng.permutation(32)
        got = engine.mfma_bf16_probe(a[:, :, perm], b[:, perm, :], c).astype(np.float64)
        worst = max(worst, float((np.abs(got - want) / (big * U)).max()))
    assert worst <= 33.0, f"a slot order gives {worst:.2f} x 2^-24 of the largest term"


def test_filter_shaped_operands(engine):
    """The filter's own operand shape (three-way bf16 splits of d = 10 coordinates, two chained instructions):
    |u - exact| <= 68.7 x 2^-24 S with S = sum |x_f c_f| + |h| (kmeans_filter.h, 'accumulation')."""
    rng = np.random.default_rng(11)
    t, d = 16, 10

    def split3(v):
        parts, r = [], np.asarray(v, np.float32)
        for _ in range(3):
            bits = _bf16_bits(r)
            p = (bits.astype(np.uint32) << 16).view(np.float32)
            parts.append(bits)
            r = (r - p).astype(np.float32)
        return parts

    cc = rng.normal(size=(t, 16, d)) * 3.0            # 16 centres per tile
    xx = rng.normal(size=(t, 16, d)) * 3.0            # 16 frames per tile
    h = 0.5 * (cc ** 2).sum(-1)
    cp, xp = split3(cc), split3(xx)                   # [part][t, 16, d]
    hp = split3(-h)
    pc = [0, 1, 0, 1, 2, 0]                           # term -> part of c, part of x (kmeans_filter.h)
    px = [0, 0, 1, 1, 0, 2]
    A = np.zeros((2, t, 16, 32), np.uint16)
    B = np.zeros((2, t, 32, 16), np.uint16)
    one = _bf16_bits(np.float32(1.0))
    slot = 0
    for term in range(6):
        for f in range(d):
            m, s_ = divmod(slot, 32)
            A[m, :, :, s_] = cp[pc[term]][:, :, f]
            B[m, :, s_, :] = xp[px[term]][:, :, f]
            slot += 1
    for part in range(3):
        m, s_ = divmod(slot, 32)
        A[m, :, :, s_] = hp[part]
        B[m, :, s_, :] = one
        slot += 1
    zero = np.zeros((t, 16, 16), np.float32)
    first = engine.mfma_bf16_probe(A[0], B[0], zero)
    got = engine.mfma_bf16_probe(A[1], B[1], first).astype(np.float64)
    # exact value of what the two instructions were asked to add up
    want = np.zeros((t, 16, 16))
    for m in range(2):
        want += np.einsum("tik,tkj->tij", _bf16_val(A[m]), _bf16_val(B[m]))
    S = np.einsum("tif,tjf->tij", np.abs(cc), np.abs(xx)) + h[:, :, None]
    err = np.abs(got - want) / (S * U)
    assert float(err.max()) <= 68.7, f"chained accumulation error {float(err.max()):.1f} x 2^-24 S exceeds the filter's bound"
