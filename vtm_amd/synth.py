"""Synthetic 10-bit 4:2:0 clips and border-extended planes (SURVEY.md section 8d / Appendix E generator).

There is no network on the build or GPU boxes, so every test and bench input is generated here from a seed.
`gen_frames` is the survey's "easy" clip (global pan + a moving square + noise); `gen_frames_hard` the
ME-stress clip (two textures moving in opposite directions in a 32x32 checkerboard, sigma-10 noise).
`extend_plane` replicates the border the way a reconstructed reference picture is padded before it is
searched (reference: Picture::extendPicBorder, CommonLib/Picture.cpp; margin >= CTU + 16 + filter taps).
"""
import numpy as np


def _texture(rng, W, H, bd):
    base = rng.integers(0, 1 << bd, size=(H // 8 + 2, W // 8 + 2)).astype(np.float64)
    yy = np.arange(H) / 8.0
    xx = np.arange(W) / 8.0
    y0 = np.floor(yy).astype(int)
    x0 = np.floor(xx).astype(int)
    fy = (yy - y0)[:, None]
    fx = (xx - x0)[None, :]
    return (base[y0][:, x0] * (1 - fy) * (1 - fx) + base[y0 + 1][:, x0] * fy * (1 - fx)
            + base[y0][:, x0 + 1] * (1 - fy) * fx + base[y0 + 1][:, x0 + 1] * fy * fx)


def gen_frames(w, h, frames, seed=1234, bd=10, chroma=False):
    """Returns a list of luma planes (int16, h x w) -- or (Y, U, V) tuples when chroma=True."""
    rng = np.random.default_rng(seed)
    W, H = w + 256, h + 256
    tex = _texture(rng, W, H, bd)
    out = []
    for t in range(frames):
        dx, dy = 3 * t, 2 * t
        Y = tex[dy:dy + h, dx:dx + w].copy()
        sx, sy = (20 + 7 * t) % (w - 64), (30 + 5 * t) % (h - 64)
        Y[sy:sy + 64, sx:sx + 64] = Y[sy:sy + 64, sx:sx + 64] * 0.5 + (1 << (bd - 1)) * 0.9
        Y += rng.normal(0, 2.0, size=Y.shape)
        Y = np.clip(np.rint(Y), 0, (1 << bd) - 1).astype(np.int16)
        if not chroma:
            out.append(Y)
            continue
        U = np.clip(np.rint(tex[dy:dy + h:2, dx:dx + w:2] * 0.25 + (1 << (bd - 1)) * 0.75), 0, (1 << bd) - 1)
        V = np.clip(np.rint((1 << bd) - 1 - tex[dy:dy + h:2, dx:dx + w:2] * 0.25 - (1 << (bd - 1)) * 0.25), 0,
                    (1 << bd) - 1)
        out.append((Y, U.astype(np.int16), V.astype(np.int16)))
    return out


def gen_frames_hard(w, h, frames, seed=4321, bd=10, sigma=10.0, chroma=False):
    rng = np.random.default_rng(seed)
    W, H = w + 256, h + 256
    A = _texture(rng, W, H, bd)
    B = _texture(rng, W, H, bd)
    yy, xx = np.mgrid[0:h, 0:w]
    mask = (((yy // 32) + (xx // 32)) & 1).astype(bool)
    out = []
    for t in range(frames):
        a = A[2 * t:2 * t + h, 5 * t:5 * t + w]
        b = B[128 - 3 * t:128 - 3 * t + h, 128 - 4 * t:128 - 4 * t + w]
        Y = np.clip(np.rint(np.where(mask, a, b) + rng.normal(0, sigma, size=(h, w))), 0, (1 << bd) - 1)
        if not chroma:
            out.append(Y.astype(np.int16))
            continue
        m2 = np.where(mask, a, b)[::2, ::2]
        U = np.clip(np.rint(m2 * 0.25 + (1 << (bd - 1)) * 0.75 + rng.normal(0, sigma / 4, size=m2.shape)), 0, (1 << bd) - 1)
        V = np.clip(np.rint((1 << bd) - 1 - m2 * 0.25 - (1 << (bd - 1)) * 0.25 + rng.normal(0, sigma / 4, size=m2.shape)), 0, (1 << bd) - 1)
        out.append((Y.astype(np.int16), U.astype(np.int16), V.astype(np.int16)))
    return out


def padded_stride(w, margin, align=64):
    return (w + 2 * margin + align - 1) // align * align


def extend_plane(plane, margin=160, align=64):
    """Edge-replicated copy with `margin` samples on every side; row stride rounded up to `align` samples.
    Returns (buf, origin_offset, stride): buf is 1-D int16, sample (x, y) lives at origin_offset + y*stride + x."""
    h, w = plane.shape
    stride = padded_stride(w, margin, align)
    ext = np.pad(plane, ((margin, margin), (margin, stride - w - margin)), mode="edge")
    buf = np.ascontiguousarray(ext, dtype=np.int16).reshape(-1)
    return buf, margin * stride + margin, stride
