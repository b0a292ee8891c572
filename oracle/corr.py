"""numpy restatement of the reference's correlation lookups.  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py; PARITY UNPINNED -- no reference fixtures exist for this path).

* corr_index_forward follows /root/reference/src/correlation_kernels.cu:19-70 (kernel) and
  :126-155 (wrapper: output zero-initialised, dtype of the volume).
* altcorr_forward follows /root/reference/src/altcorr_kernel.cu:27-149 (kernel) and :290-319.
"""
from __future__ import annotations

import numpy as np


def corr_index_forward(volume, coords, radius):
    """volume [B,H1,W1,H2,W2] (f16/f32/f64), coords [B,2,H1,W1] f32 -> corr [B,2r+1,2r+1,H1,W1].

    Arithmetic is done in the volume's dtype exactly as the kernel does (ck:55-65): each bilinear
    weight is formed in fp32, rounded to scalar_t, the product is rounded to scalar_t and the
    running sum is rounded to scalar_t after every add.  For f16 numpy's half ops (compute in
    fp32, round to half) match c10::Half's operators.  Out-of-range taps are skipped (ck:52).
    corr[n][i][j] has dim1 = x offset, dim2 = y offset (ck:57-66).
    """
    volume = np.asarray(volume)
    st = volume.dtype
    coords = np.asarray(coords, dtype=np.float32)
    B, H1, W1, H2, W2 = volume.shape
    r = int(radius)
    rd = 2 * r + 1
    x0 = coords[:, 0]
    y0 = coords[:, 1]
    fx = np.floor(x0)
    fy = np.floor(y0)
    dx = (x0 - fx).astype(np.float32)  # ck:42-43
    dy = (y0 - fy).astype(np.float32)
    one = np.float32(1.0)
    w_se = (dx * dy).astype(st)               # tap (i,j) -> out (i-1,j-1)   ck:55-56
    w_sw = (dx * (one - dy)).astype(st)       # tap (i,j) -> out (i-1,j)     ck:58-59
    w_ne = ((one - dx) * dy).astype(st)       # tap (i,j) -> out (i,j-1)     ck:61-62
    w_nw = ((one - dx) * (one - dy)).astype(st)  # tap (i,j) -> out (i,j)    ck:64-65
    with np.errstate(invalid="ignore"):
        fxi = np.where(np.isfinite(fx), fx, -1e6).astype(np.int64)
        fyi = np.where(np.isfinite(fy), fy, -1e6).astype(np.int64)
    corr = np.zeros((B, rd, rd, H1, W1), dtype=st)
    bb, yy, xx = np.meshgrid(np.arange(B), np.arange(H1), np.arange(W1), indexing="ij")
    for i in range(rd + 1):
        for j in range(rd + 1):
            x1 = fxi - r + i
            y1 = fyi - r + j
            inb = (x1 >= 0) & (x1 < W2) & (y1 >= 0) & (y1 < H2)
            s = volume[bb, yy, xx, np.clip(y1, 0, H2 - 1), np.clip(x1, 0, W2 - 1)]
            if i > 0 and j > 0:
                upd = (corr[:, i - 1, j - 1] + (s * w_se).astype(st)).astype(st)
                corr[:, i - 1, j - 1] = np.where(inb, upd, corr[:, i - 1, j - 1])
            if i > 0 and j < rd:
                upd = (corr[:, i - 1, j] + (s * w_sw).astype(st)).astype(st)
                corr[:, i - 1, j] = np.where(inb, upd, corr[:, i - 1, j])
            if i < rd and j > 0:
                upd = (corr[:, i, j - 1] + (s * w_ne).astype(st)).astype(st)
                corr[:, i, j - 1] = np.where(inb, upd, corr[:, i, j - 1])
            if i < rd and j < rd:
                upd = (corr[:, i, j] + (s * w_nw).astype(st)).astype(st)
                corr[:, i, j] = np.where(inb, upd, corr[:, i, j])
    return corr


def corr_index_backward(volume_shape, coords, corr_grad, radius):
    """Gradient of `corr_index_forward` with respect to the volume, restating the scatter of
    /root/reference/src/correlation_kernels.cu:73-124 (wrapper :157-185: zero-initialised, dtype of the
    volume): for every query and each of the (2r+2)^2 integer taps inside the plane, g accumulates up to
    four products corr_grad * scalar_t(weight) in scalar_t, in the order (i-1,j-1)*dx*dy, (i-1,j)*dx*(1-dy),
    (i,j-1)*(1-dx)*dy, (i,j)*(1-dx)*(1-dy) (ck:106-117), and is added to volume_grad[n][y][x][y1][x1]
    (each query owns its plane: one add per element onto zero)."""
    cg = np.asarray(corr_grad)
    st = cg.dtype
    coords = np.asarray(coords, dtype=np.float32)
    B, H1, W1, H2, W2 = volume_shape
    r = int(radius)
    rd = 2 * r + 1
    x0, y0 = coords[:, 0], coords[:, 1]
    fx, fy = np.floor(x0), np.floor(y0)
    dx = (x0 - fx).astype(np.float32)
    dy = (y0 - fy).astype(np.float32)
    one = np.float32(1.0)
    w11 = (dx * dy).astype(st)
    w10 = (dx * (one - dy)).astype(st)
    w01 = ((one - dx) * dy).astype(st)
    w00 = ((one - dx) * (one - dy)).astype(st)
    with np.errstate(invalid="ignore"):
        fxi = np.where(np.isfinite(fx), fx, -1e6).astype(np.int64)
        fyi = np.where(np.isfinite(fy), fy, -1e6).astype(np.int64)
    grad = np.zeros((B, H1, W1, H2, W2), dtype=st)
    bb, yy, xx = np.meshgrid(np.arange(B), np.arange(H1), np.arange(W1), indexing="ij")
    for i in range(rd + 1):
        for j in range(rd + 1):
            x1 = fxi - r + i
            y1 = fyi - r + j
            inb = (x1 >= 0) & (x1 < W2) & (y1 >= 0) & (y1 < H2)
            g = np.zeros((B, H1, W1), dtype=st)
            if i > 0 and j > 0:
                g = (g + (cg[:, i - 1, j - 1] * w11).astype(st)).astype(st)
            if i > 0 and j < rd:
                g = (g + (cg[:, i - 1, j] * w10).astype(st)).astype(st)
            if i < rd and j > 0:
                g = (g + (cg[:, i, j - 1] * w01).astype(st)).astype(st)
            if i < rd and j < rd:
                g = (g + (cg[:, i, j] * w00).astype(st)).astype(st)
            grad[bb[inb], yy[inb], xx[inb], y1[inb], x1[inb]] = g[inb]
    return grad


def altcorr_forward(fmap1, fmap2, coords, radius, acc_dtype=None, chunked=True):
    """fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C], coords [B,N,H1,W1,2] f32 -> corr [B,N,(2r+1)^2,H1,W1].

    Channel of output = ix*(2r+1) + iy (x-major; ak:109-112).  ``acc_dtype`` is the arithmetic
    type (default: fmap1's dtype, like the kernel's scalar_t; pass np.float64 for the "truth").
    With ``chunked`` the dot product is formed in 32-channel chunks, each chunk scattered to the
    four bilinear neighbours before the next one (ak:52, :98-142); otherwise as one dot product.
    """
    fmap1 = np.asarray(fmap1)
    fmap2 = np.asarray(fmap2)
    st = np.dtype(acc_dtype) if acc_dtype is not None else fmap1.dtype
    coords = np.asarray(coords, dtype=np.float32)
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    N = coords.shape[1]
    r = int(radius)
    rd = 2 * r + 1
    f1 = fmap1.astype(st)
    f2 = fmap2.astype(st)
    corr = np.zeros((B, N, rd * rd, H1, W1), dtype=st)
    one = np.float32(1.0)
    CH = 32 if chunked else C  # CHANNEL_STRIDE ak:19
    bb = np.arange(B)[:, None, None]
    for c0 in range(0, C, CH):
        f1c = f1[..., c0:c0 + CH]
        for n in range(N):
            x2 = coords[:, n, :, :, 0]
            y2 = coords[:, n, :, :, 1]
            fx = np.floor(x2)
            fy = np.floor(y2)
            dx = (x2 - fx).astype(np.float32)  # ak:78-79
            dy = (y2 - fy).astype(np.float32)
            fxi = fx.astype(np.int64)
            fyi = fy.astype(np.int64)
            w_nw = (dy * dx).astype(st)                    # ak:119
            w_ne = (dy * (one - dx)).astype(st)            # ak:120
            w_sw = ((one - dy) * dx).astype(st)            # ak:121
            w_se = ((one - dy) * (one - dx)).astype(st)    # ak:122
            for iy in range(rd + 1):
                for ix in range(rd + 1):
                    h2 = fyi - r + iy
                    w2 = fxi - r + ix
                    inb = (h2 >= 0) & (h2 < H2) & (w2 >= 0) & (w2 < W2)
                    g = f2[bb, np.clip(h2, 0, H2 - 1), np.clip(w2, 0, W2 - 1), c0:c0 + CH]
                    g = np.where(inb[..., None], g, np.zeros((), st))  # ak:90-94
                    if st == np.float64 or not chunked:
                        s = np.einsum("bhwc,bhwc->bhw", f1c, g).astype(st)
                    else:
                        s = np.zeros((B, H1, W1), dtype=st)
                        for k in range(f1c.shape[-1]):  # ak:98-100, sequential in scalar_t
                            s = (s + (f1c[..., k] * g[..., k]).astype(st)).astype(st)
                    if iy > 0 and ix > 0:
                        ch = (iy - 1) + rd * (ix - 1)
                        corr[:, n, ch] = (corr[:, n, ch] + (s * w_nw).astype(st)).astype(st)
                    if iy > 0 and ix < rd:
                        ch = (iy - 1) + rd * ix
                        corr[:, n, ch] = (corr[:, n, ch] + (s * w_ne).astype(st)).astype(st)
                    if iy < rd and ix > 0:
                        ch = iy + rd * (ix - 1)
                        corr[:, n, ch] = (corr[:, n, ch] + (s * w_sw).astype(st)).astype(st)
                    if iy < rd and ix < rd:
                        ch = iy + rd * ix
                        corr[:, n, ch] = (corr[:, n, ch] + (s * w_se).astype(st)).astype(st)
    return corr


def altcorr_backward(fmap1, fmap2, coords, corr_grad, radius):
    """Gradients of `altcorr_forward` with respect to the two feature maps (fp64), restating the scatter of
    altcorr_kernel.cu:152-286: for every query and every one of the (2r+2)^2 integer taps the four bilinear
    weights combine the matching entries of corr_grad into one scalar g (ak:228-246), then
    fmap1_grad[query] += g * fmap2[tap] and fmap2_grad[tap] += g * fmap1[query] (taps outside fmap2 are skipped).
    The coordinate gradient of the reference is identically zero (ak:322-355 returns zeros)."""
    f1 = np.asarray(fmap1, np.float64)
    f2 = np.asarray(fmap2, np.float64)
    cg = np.asarray(corr_grad, np.float64)
    coords = np.asarray(coords, dtype=np.float32)
    B, H1, W1, C = f1.shape
    _, H2, W2, _ = f2.shape
    N = coords.shape[1]
    r = int(radius)
    rd = 2 * r + 1
    g1 = np.zeros_like(f1)
    g2 = np.zeros_like(f2)
    bb = np.broadcast_to(np.arange(B)[:, None, None], (B, H1, W1))
    one = np.float32(1.0)
    for n in range(N):
        x2, y2 = coords[:, n, :, :, 0], coords[:, n, :, :, 1]
        fx, fy = np.floor(x2), np.floor(y2)
        dx = (x2 - fx).astype(np.float32)
        dy = (y2 - fy).astype(np.float32)
        fxi, fyi = fx.astype(np.int64), fy.astype(np.int64)
        # the four bilinear weights as the forward forms them: fp32 products (ak:119-122)
        w_nw, w_ne = (dy * dx).astype(np.float64), (dy * (one - dx)).astype(np.float64)
        w_sw, w_se = ((one - dy) * dx).astype(np.float64), ((one - dy) * (one - dx)).astype(np.float64)
        for iy in range(rd + 1):
            for ix in range(rd + 1):
                h2, w2 = fyi - r + iy, fxi - r + ix
                inb = (h2 >= 0) & (h2 < H2) & (w2 >= 0) & (w2 < W2)
                g = np.zeros((B, H1, W1))
                if iy > 0 and ix > 0:
                    g += cg[:, n, (iy - 1) + rd * (ix - 1)] * w_nw
                if iy > 0 and ix < rd:
                    g += cg[:, n, (iy - 1) + rd * ix] * w_ne
                if iy < rd and ix > 0:
                    g += cg[:, n, iy + rd * (ix - 1)] * w_sw
                if iy < rd and ix < rd:
                    g += cg[:, n, iy + rd * ix] * w_se
                g = np.where(inb, g, 0.0)
                hc, wc = np.clip(h2, 0, H2 - 1), np.clip(w2, 0, W2 - 1)
                g1 += g[..., None] * f2[bb, hc, wc]
                np.add.at(g2, (bb, hc, wc), g[..., None] * f1)
    return g1, g2
