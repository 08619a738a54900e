#!/usr/bin/env python3
"""Development aid (CPU, numpy): rounding error of Winograd F(2x2,3x3) and F(4x4,3x3) evaluated in fp32 against a float64
direct convolution, Cin = 64, unit-variance data and He-scaled weights -- what a move to the larger tile would cost in
accuracy.  Transforms in fp32, channel sums in fp32 (pairwise, as numpy sums)."""
import numpy as np

rng = np.random.default_rng(0)


def mats(m):
    if m == 2:
        BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
        G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
        AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
    else:
        BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                       [0, 4, 0, -5, 0, 1]], np.float64)
        G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                      [0, 0, 1]], np.float64)
        AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)
    return BT, G, AT


def winograd(x, w, m, dt=np.float32):
    """x (C, H, W), w (K, C, 3, 3) -> (K, H-2, W-2), 'valid' correlation, tiles of m x m outputs."""
    BT, G, AT = mats(m)
    t = m + 2
    C, H, W = x.shape
    K = w.shape[0]
    U = np.einsum("ia,kcab,jb->kcij", G, w.astype(np.float64), G).astype(dt)        # rounded once, as the pack kernel does
    ty, tx = (H - 2) // m, (W - 2) // m
    y = np.zeros((K, ty * m, tx * m), dt)
    BTd, ATd = BT.astype(dt), AT.astype(dt)
    for i in range(ty):
        for j in range(tx):
            d = x[:, i * m:i * m + t, j * m:j * m + t].astype(dt)
            V = np.einsum("ia,cab,jb->cij", BTd, d, BTd).astype(dt)
            M = np.einsum("kcij,cij->kij", U, V).astype(dt)
            y[:, i * m:(i + 1) * m, j * m:(j + 1) * m] = np.einsum("ia,kab,jb->kij", ATd, M, ATd).astype(dt)
    return y


def direct(x, w):
    C, H, W = x.shape
    K = w.shape[0]
    y = np.zeros((K, H - 2, W - 2))
    for a in range(3):
        for b in range(3):
            y += np.einsum("kc,chw->khw", w[:, :, a, b].astype(np.float64), x[:, a:a + H - 2, b:b + W - 2].astype(np.float64))
    return y


C, K, H, W = 64, 32, 26, 26
x = rng.standard_normal((C, H, W)).astype(np.float32)
w = (rng.standard_normal((K, C, 3, 3)) * np.sqrt(2.0 / (C * 9))).astype(np.float32)
ref = direct(x, w)
d32 = np.zeros_like(ref, dtype=np.float32)
for a in range(3):
    for b in range(3):
        d32 += np.einsum("kc,chw->khw", w[:, :, a, b], x[:, a:a + H - 2, b:b + W - 2]).astype(np.float32)
rms = np.sqrt((ref ** 2).mean())
print(f"output rms {rms:.3f}")
print(f"direct fp32        : max err {np.abs(d32 - ref).max() / rms:.2e}  rms err {np.sqrt(((d32 - ref) ** 2).mean()) / rms:.2e}  (relative to the output rms)")
for m in (2, 4):
    y = winograd(x, w, m)
    r = ref[:, :y.shape[1], :y.shape[2]]
    print(f"F({m}x{m},3x3) fp32    : max err {np.abs(y - r).max() / rms:.2e}  rms err {np.sqrt(((y - r) ** 2).mean()) / rms:.2e}")
