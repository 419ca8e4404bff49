"""The handful of OpenCV calls the reference's result section makes (`fsdp_chunked_coherent.py:236-253`), on numpy:

    cvtColor(frame, COLOR_BGR2GRAY | COLOR_RGB2BGR | ...)              8-bit, OpenCV's fixed-point grey weights
    calcOpticalFlowFarneback(prev, next, None, 0.5, 3, 15, 3, 5, 1.2, 0)   dense flow (h, w, 2) float32
    remap(src, map_x, map_y, INTER_LINEAR)                             bilinear, constant-0 border
    VideoWriter(path, VideoWriter_fourcc(*"mp4v"), fps, (w, h)).write(bgr) / .release()

`calcOpticalFlowFarneback` is an independent implementation of the published algorithm (G. Farneback, "Two-frame
motion estimation based on polynomial expansion", SCIA 2003) with OpenCV's parameterisation: Gaussian image pyramid,
quadratic polynomial expansion under a Gaussian applicability of radius `poly_n`, `iterations` rounds of
displacement estimation with a `winsize` box window per level.  OpenCV itself is not available in this image, so
the agreement of `flow_err` with the reference's own number is NOT pinned by any fixture ("parity unpinned");
tests pin the implementation on synthetic translations instead.

`VideoWriter` writes Motion-JPEG samples (Pillow encodes them) into an ISO base-media (.mp4) file under the `mp4v`
sample entry with object type 0x6C — the container and fourcc the reference asks for; an MPEG-4 part 2 encoder does
not exist here.
"""
from __future__ import annotations

import io
import struct

import numpy as np
from scipy import ndimage

COLOR_BGR2GRAY, COLOR_RGB2GRAY, COLOR_RGB2BGR, COLOR_BGR2RGB = 6, 7, 4, 4
COLOR_GRAY2BGR = 8
INTER_NEAREST, INTER_LINEAR = 0, 1
OPTFLOW_USE_INITIAL_FLOW, OPTFLOW_FARNEBACK_GAUSSIAN = 4, 256


class error(Exception):
    pass


# ---------------------------------------------------------------------------------------------
def cvtColor(src, code):
    src = np.asarray(src)
    if code in (COLOR_BGR2GRAY, COLOR_RGB2GRAY):
        if src.ndim != 3 or src.shape[2] < 3:
            raise error("cvtColor: expected a 3-channel image")
        c0, c1, c2 = (src[..., i].astype(np.int64) for i in range(3))
        b, g, r = (c0, c1, c2) if code == COLOR_BGR2GRAY else (c2, c1, c0)
        if src.dtype == np.uint8:                         # OpenCV: 14-bit fixed point, round to nearest
            return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)
        return (0.114 * b + 0.587 * g + 0.299 * r).astype(src.dtype)
    if code == COLOR_RGB2BGR:                             # (== COLOR_BGR2RGB)
        return np.ascontiguousarray(src[..., ::-1])
    if code == COLOR_GRAY2BGR:
        return np.repeat(src[..., None], 3, axis=2)
    raise error(f"cvtColor: conversion code {code} is not provided by this shim")


def _resize_linear(img, w, h):
    """cv2.resize(..., INTER_LINEAR): pixel centres map as src = (dst + 0.5) * scale - 0.5, replicated border."""
    H, W = img.shape[:2]
    ys = np.clip((np.arange(h) + 0.5) * (H / h) - 0.5, 0, H - 1)
    xs = np.clip((np.arange(w) + 0.5) * (W / w) - 0.5, 0, W - 1)
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    y1, x1 = np.minimum(y0 + 1, H - 1), np.minimum(x0 + 1, W - 1)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    if img.ndim == 3:
        fy, fx = fy[..., None], fx[..., None]
    a, b = img[y0][:, x0], img[y0][:, x1]
    c, d = img[y1][:, x0], img[y1][:, x1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def remap(src, map1, map2, interpolation=INTER_LINEAR, borderValue=0):
    src = np.asarray(src)
    H, W = src.shape[:2]
    x, y = np.asarray(map1, np.float64), np.asarray(map2, np.float64)
    if interpolation == INTER_NEAREST:
        xi, yi = np.rint(x).astype(int), np.rint(y).astype(int)
        ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
        out = np.full(x.shape + src.shape[2:], borderValue, dtype=src.dtype)
        out[ok] = src[yi[ok], xi[ok]]
        return out
    x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
    fx, fy = x - x0, y - y0
    acc = np.zeros(x.shape + src.shape[2:], np.float64)

    def tap(yy, xx, wgt):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = np.where(ok[(...,) + (None,) * (src.ndim - 2)], src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], borderValue)
        return v * wgt[(...,) + (None,) * (src.ndim - 2)]

    acc += tap(y0, x0, (1 - fx) * (1 - fy)) + tap(y0, x0 + 1, fx * (1 - fy))
    acc += tap(y0 + 1, x0, (1 - fx) * fy) + tap(y0 + 1, x0 + 1, fx * fy)
    if np.issubdtype(src.dtype, np.integer):
        return np.clip(np.rint(acc), np.iinfo(src.dtype).min, np.iinfo(src.dtype).max).astype(src.dtype)
    return acc.astype(src.dtype)


# ---------------------------------------------------------------------------------------------
# Farneback dense optical flow
# ---------------------------------------------------------------------------------------------
def _poly_exp(img, n, sigma):
    """Quadratic polynomial expansion f(x0 + d) ~ d^T A d + b^T d + c under a Gaussian applicability of radius n:
    weighted least squares on the basis (1, x, y, x^2, y^2, xy).  Returns bx, by, axx, ayy, axy (each (h, w))."""
    x = np.arange(-n, n + 1, dtype=np.float64)
    g = np.exp(-x * x / (2 * sigma * sigma))
    g /= g.sum()
    k0, k1, k2 = g, g * x, g * x * x

    def sep(ky, kx):                                       # correlation with ky (rows) x kx (columns), reflect-101 border
        t = ndimage.correlate1d(img, kx, axis=1, mode="mirror")
        return ndimage.correlate1d(t, ky, axis=0, mode="mirror")

    m = np.stack([sep(k0, k0), sep(k0, k1), sep(k1, k0), sep(k0, k2), sep(k2, k0), sep(k1, k1)], -1)   # 1,x,y,xx,yy,xy
    # Gram matrix of the basis under the applicability a(x, y) = g(x) g(y)
    X, Y = np.meshgrid(x, x)
    a = np.outer(g, g)
    basis = np.stack([np.ones_like(X), X, Y, X * X, Y * Y, X * Y], 0).reshape(6, -1)
    G = (basis * a.reshape(1, -1)) @ basis.T
    r = m @ np.linalg.inv(G).T                             # coefficients (c, bx, by, axx, ayy, axy)
    return r[..., 1], r[..., 2], r[..., 3], r[..., 4], r[..., 5]


def _sample(img, x, y):
    """Bilinear sample with the coordinates clamped to the image (displaced neighbourhoods near the border)."""
    H, W = img.shape
    x = np.clip(x, 0, W - 1)
    y = np.clip(y, 0, H - 1)
    x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
    x1, y1 = np.minimum(x0 + 1, W - 1), np.minimum(y0 + 1, H - 1)
    fx, fy = x - x0, y - y0
    return (img[y0, x0] * (1 - fx) + img[y0, x1] * fx) * (1 - fy) + (img[y1, x0] * (1 - fx) + img[y1, x1] * fx) * fy


def _update_flow(R0, R1, flow, winsize, gaussian):
    """One displacement update: A = (A0 + A1(x + d)) / 2, db = -(b1(x + d) - b0) / 2 + A d, then
    d = (sum w A^T A)^-1 (sum w A^T db) over the window."""
    h, w = flow.shape[:2]
    gx, gy = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    xs, ys = gx + flow[..., 0], gy + flow[..., 1]
    bx1, by1, axx1, ayy1, axy1 = (_sample(c, xs, ys) for c in R1)
    bx0, by0, axx0, ayy0, axy0 = R0
    a11, a22, a12 = 0.5 * (axx0 + axx1), 0.5 * (ayy0 + ayy1), 0.25 * (axy0 + axy1)     # A = [[a11, a12], [a12, a22]]
    dbx = -0.5 * (bx1 - bx0) + a11 * flow[..., 0] + a12 * flow[..., 1]
    dby = -0.5 * (by1 - by0) + a12 * flow[..., 0] + a22 * flow[..., 1]
    comps = [a11 * a11 + a12 * a12, a11 * a12 + a12 * a22, a12 * a12 + a22 * a22, a11 * dbx + a12 * dby, a12 * dbx + a22 * dby]
    if gaussian:
        blur = lambda c: ndimage.gaussian_filter(c, winsize * 0.3, mode="mirror", truncate=(winsize // 2) / (winsize * 0.3))  # noqa: E731
    else:
        blur = lambda c: ndimage.uniform_filter(c, winsize, mode="mirror")   # noqa: E731
    g11, g12, g22, h1, h2 = (blur(c) for c in comps)
    det = g11 * g22 - g12 * g12 + 1e-3                     # (OpenCV regularises the same way)
    out = np.empty_like(flow)
    out[..., 0] = (g22 * h1 - g12 * h2) / det
    out[..., 1] = (g11 * h2 - g12 * h1) / det
    return out


def calcOpticalFlowFarneback(prev, next, flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags):
    p0, p1 = np.asarray(prev, np.float64), np.asarray(next, np.float64)
    if p0.ndim != 2 or p0.shape != p1.shape:
        raise error("calcOpticalFlowFarneback: two single-channel images of equal size expected")
    H, W = p0.shape
    levels = max(int(levels), 1)
    while levels > 1 and min(H, W) * pyr_scale ** (levels - 1) < 16:       # (OpenCV stops at levels that get too small)
        levels -= 1
    use_init = bool(flags & OPTFLOW_USE_INITIAL_FLOW) and flow is not None
    cur = None
    for k in range(levels - 1, -1, -1):
        scale = pyr_scale ** k
        w, h = max(int(round(W * scale)), 1), max(int(round(H * scale)), 1)
        if k > 0:
            sigma = (1.0 / scale - 1.0) * 0.5
            i0 = _resize_linear(ndimage.gaussian_filter(p0, sigma, mode="mirror"), w, h)
            i1 = _resize_linear(ndimage.gaussian_filter(p1, sigma, mode="mirror"), w, h)
        else:
            i0, i1 = p0, p1
        if cur is None:
            cur = _resize_linear(np.asarray(flow, np.float64), w, h) * scale if use_init else np.zeros((h, w, 2))
        else:
            cur = _resize_linear(cur, w, h) / pyr_scale
        R0, R1 = _poly_exp(i0, poly_n, poly_sigma), _poly_exp(i1, poly_n, poly_sigma)
        for _ in range(max(int(iterations), 1)):
            cur = _update_flow(R0, R1, cur, winsize, bool(flags & OPTFLOW_FARNEBACK_GAUSSIAN))
    return cur.astype(np.float32)


# ---------------------------------------------------------------------------------------------
# VideoWriter: Motion-JPEG in an .mp4 (ISO BMFF) container
# ---------------------------------------------------------------------------------------------
def VideoWriter_fourcc(c1, c2, c3, c4):
    return (ord(c1) & 255) | ((ord(c2) & 255) << 8) | ((ord(c3) & 255) << 16) | ((ord(c4) & 255) << 24)


def _box(kind: bytes, payload: bytes) -> bytes:
    return struct.pack(">I", 8 + len(payload)) + kind + payload


def _full(kind: bytes, version: int, flags: int, payload: bytes) -> bytes:
    return _box(kind, struct.pack(">I", (version << 24) | flags) + payload)


class VideoWriter:
    def __init__(self, filename=None, fourcc=0, fps=0.0, frameSize=(0, 0), isColor=True):
        self._f = None
        if filename is not None:
            self.open(filename, fourcc, fps, frameSize, isColor)

    def open(self, filename, fourcc, fps, frameSize, isColor=True):
        self._path, self._fps, (self._w, self._h) = filename, float(fps), (int(frameSize[0]), int(frameSize[1]))
        self._sizes = []
        self._f = open(filename, "wb")
        self._f.write(_box(b"ftyp", b"isom" + struct.pack(">I", 0x200) + b"isomiso2mp41"))
        self._mdat_pos = self._f.tell()
        self._f.write(struct.pack(">I", 0) + b"mdat")          # size patched in release()
        return True

    def isOpened(self):
        return self._f is not None

    def write(self, frame):
        from PIL import Image
        frame = np.asarray(frame)
        if frame.shape[:2] != (self._h, self._w):
            raise error(f"VideoWriter.write: frame {frame.shape[1]}x{frame.shape[0]} != {self._w}x{self._h}")
        rgb = frame[..., ::-1] if frame.ndim == 3 else frame       # OpenCV frames are BGR
        buf = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(rgb)).save(buf, format="JPEG", quality=92)
        data = buf.getvalue()
        self._f.write(data)
        self._sizes.append(len(data))

    def release(self):
        if self._f is None:
            return
        f, n = self._f, len(self._sizes)
        end = f.tell()
        f.seek(self._mdat_pos)
        f.write(struct.pack(">I", end - self._mdat_pos))
        f.seek(end)
        ts = max(int(round(self._fps * 1000)), 1)                  # media timescale; one sample lasts 1000 ticks
        dur = n * 1000
        mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, ts, dur) + struct.pack(">IH", 0x00010000, 0x0100) + b"\0" * 10 +
                     struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000) + b"\0" * 24 + struct.pack(">I", 2))
        tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, dur) + b"\0" * 8 + struct.pack(">HHHH", 0, 0, 0, 0) +
                     struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000) +
                     struct.pack(">II", self._w << 16, self._h << 16))
        mdhd = _full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, ts, dur, 0x55C4, 0))
        hdlr = _full(b"hdlr", 0, 0, struct.pack(">I", 0) + b"vide" + b"\0" * 12 + b"VideoHandler\0")
        # ES descriptor: object type 0x6C = JPEG (ISO/IEC 10918-1), stream type 4 = visual
        dcd = bytes([0x04, 13, 0x6C, 0x11, 0, 0, 0]) + struct.pack(">II", 0, 0)
        esd = bytes([0x03, 3 + len(dcd) + 3, 0, 1, 0]) + dcd + bytes([0x06, 1, 2])
        entry = (b"\0" * 6 + struct.pack(">H", 1) + b"\0" * 16 + struct.pack(">HH", self._w, self._h) +
                 struct.pack(">II", 0x00480000, 0x00480000) + struct.pack(">I", 0) + struct.pack(">H", 1) + b"\0" * 32 +
                 struct.pack(">Hh", 24, -1) + _full(b"esds", 0, 0, esd))
        stsd = _full(b"stsd", 0, 0, struct.pack(">I", 1) + _box(b"mp4v", entry))
        stts = _full(b"stts", 0, 0, struct.pack(">III", 1, n, 1000))
        stsc = _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, max(n, 1), 1))
        stsz = _full(b"stsz", 0, 0, struct.pack(">II", 0, n) + b"".join(struct.pack(">I", s) for s in self._sizes))
        stco = _full(b"stco", 0, 0, struct.pack(">II", 1, self._mdat_pos + 8))
        stbl = _box(b"stbl", stsd + stts + stsc + stsz + stco)
        dinf = _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1) + _full(b"url ", 0, 1, b"")))
        minf = _box(b"minf", _full(b"vmhd", 0, 1, b"\0" * 8) + dinf + stbl)
        trak = _box(b"trak", tkhd + _box(b"mdia", mdhd + hdlr + minf))
        f.write(_box(b"moov", mvhd + trak))
        f.close()
        self._f = None
