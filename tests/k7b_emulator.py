"""Test infrastructure: a lane-level NumPy walk through csrc/tattn2.hip (K7, second design) for ONE wave.

It executes the kernel's own addressing — the unit / tile / slot arithmetic of the weight ring, the MFMA operand and
accumulator lane maps of gfx950 (`v_mfma_f32_16x16x32_f16`, `v_mfma_f32_16x16x16_f16`: /opt/skills/guides
cdna_hip_programming.md §3), the accumulator-as-operand hand-offs, the permuted k index of the output projection —
on the blob `packing.pack_k7b` produces, so that the host packing and the device indexing are checked against each
other without a GPU.  Arithmetic is fp32 with fp16 roundings where the kernel rounds.  Not product code.
"""
from __future__ import annotations

import numpy as np

UB, NU = 8192, 5


def _mfma(a, b, c, kq):
    """a, b: [64][kq*?]: lane l holds A[l & 15][kq*(l >> 4) + e], B[kq*(l >> 4) + e][l & 15]; c: [64][4] with
    C[4*(l >> 4) + r][l & 15].  kq = 8 (16x16x32) or 4 (16x16x16)."""
    lane = np.arange(64)
    n16, q4 = lane & 15, lane >> 4
    K = 4 * kq
    A = np.zeros((16, K), np.float32)
    Bm = np.zeros((K, 16), np.float32)
    for e in range(kq):
        A[n16, kq * q4 + e] = a[:, e]
        Bm[kq * q4 + e, n16] = b[:, e]
    D = A @ Bm
    out = np.array(c, np.float32, copy=True)
    for r in range(4):
        out[:, r] += D[4 * q4 + r, n16]
    return out


def mfma32(a, b, c):
    return _mfma(a, b, c, 8)


def mfma16(a, b, c):
    return _mfma(a, b, c, 4)


def h16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


class Wave:
    """One wave of tattn2_kernel<320> on its 48-row group.  `xn`: the wave's rows after P0 ([48][inner], fp16 values)."""

    def __init__(self, blob_f16: np.ndarray, inner: int, F: int, rot: int):
        assert inner == 320
        self.inner, self.F, self.rot = inner, F, rot
        self.heads, self.km = inner // 64, inner // 64
        self.uph = 3 * self.km
        self.ncgf = inner // 128
        raw = blob_f16.view(np.uint8)
        nunits = self.uph * self.heads + 2 * self.heads * self.ncgf + self.heads
        self.wqkv = raw[: self.uph * self.heads * UB]
        self.wo = raw[self.uph * self.heads * UB: nunits * UB]
        vec = raw[nunits * UB:].view(np.float32)
        self.bq, self.bo2 = vec[:inner], vec[inner:2 * inner]
        lane = np.arange(64)
        self.n16, self.q4 = lane & 15, lane >> 4
        g = np.array([0, 2, 3, 1])[self.n16 >> 2]
        self.woff = self.n16 * 64 + ((self.q4 ^ g) << 4)

    # ---- kernel addressing
    def unit_bytes(self, u):
        """unit_src<U>() of the kernel: the 8 KB the DMA copies into ring slot u % NU."""
        H = self.heads
        if u < self.uph * H:
            hs, w = divmod(u, self.uph)
            h = (hs + self.rot) % H
            off = (h * self.uph + w) * UB
            return self.wqkv[off:off + UB]
        v = u - self.uph * H
        if v < 2 * H * self.ncgf:
            cg, r = divmod(v, 2 * H)
            hs, kk = divmod(r, 2)
            h = (hs + self.rot) % H
            off = (cg * 2 * H + h * 2 + kk) * UB
        else:
            hs = v - 2 * H * self.ncgf
            h = (hs + self.rot) % H
            off = (2 * H * self.ncgf + h) * UB
        return self.wo[off:off + UB]

    def wfrag(self, u, tile):
        unit = self.unit_bytes(u)
        out = np.zeros((64, 8), np.float32)
        for l in range(64):
            o = tile * 1024 + self.woff[l]
            out[l] = unit[o:o + 16].view(np.float16).astype(np.float32)
        return out

    def xfrag(self, xn, i, ks):
        out = np.zeros((64, 8), np.float32)
        for l in range(64):
            row = 16 * i + self.n16[l]
            c0 = 8 * (4 * ks + self.q4[l])
            out[l] = xn[row, c0:c0 + 8]
        return out

    def ub(self, s):
        KM, H = self.km, self.heads
        hsteps, p1s = 2 * KM, 2 * KM * H
        if s <= p1s:
            hs, r = divmod(s, hsteps)
            return self.uph * hs + (2 * r if r < KM else 2 * KM + (r - KM))
        v = s - p1s
        c, m = divmod(v, H)
        return self.uph * H + (2 * H * c + 2 * m if c < self.ncgf else 2 * H * self.ncgf + (v - self.ncgf * H))

    # ---- the tile
    def run(self, xn, resid):
        """-> [48][inner] output rows (fp16 values)."""
        KM, H, F = self.km, self.heads, self.F
        n16, q4 = self.n16, self.q4
        fmagic = (65536 + F - 1) // F
        qpix = [((16 * i + n16) * fmagic) >> 16 for i in range(3)]
        kpix = [[((16 * i + 4 * q4 + e) * fmagic) >> 16 for e in range(4)] for i in range(3)]
        need = 0
        for a in range(3):
            for b in range(3):
                alo, ahi, blo, bhi = 16 * a // F, (16 * a + 15) // F, 16 * b // F, (16 * b + 15) // F
                if not (ahi < blo or bhi < alo):
                    need |= 1 << (3 * a + b)
                if alo == ahi and blo == bhi and alo == blo:
                    need |= 1 << (9 + 3 * a + b)
        oh = [[[None, None] for _ in range(3)] for _ in range(H)]
        z = np.zeros((64, 4), np.float32)
        for hs in range(H):
            h = (hs + self.rot) % H
            bqv = [np.stack([self.bq[h * 64 + 16 * j + 4 * q4 + e] for e in range(4)], 1) for j in range(4)]
            aq = [[bqv[j].copy() for j in range(4)] for _ in range(3)]
            ak = [[z.copy() for _ in range(4)] for _ in range(3)]
            av = [[z.copy() for _ in range(4)] for _ in range(3)]
            for m in range(KM):
                s = hs * 2 * KM + m
                u0 = self.ub(s)
                for kk in range(2):
                    wq = [self.wfrag(u0, 4 * kk + j) for j in range(4)]
                    wk = [self.wfrag(u0 + 1, 4 * kk + j) for j in range(4)]
                    x = [self.xfrag(xn, i, 2 * m + kk) for i in range(3)]
                    for j in range(4):
                        for i in range(3):
                            aq[i][j] = mfma32(wq[j], x[i], aq[i][j])
                            ak[i][j] = mfma32(wk[j], x[i], ak[i][j])
            # scores
            qh = [[h16(aq[i][j]) for j in range(4)] for i in range(3)]
            kh = [[h16(ak[i][j]) for j in range(4)] for i in range(3)]
            pt = [[None] * 3 for _ in range(3)]
            for qt in range(3):
                sc = []
                for kt in range(3):
                    c = z.copy()
                    if (need >> (3 * qt + kt)) & 1:
                        for j in range(4):
                            c = mfma16(kh[kt][j], qh[qt][j], c)
                    sc.append(c)
                ok = np.zeros((3, 64, 4), bool)
                for kt in range(3):
                    pure = (need >> (9 + 3 * qt + kt)) & 1
                    nd = (need >> (3 * qt + kt)) & 1
                    for e in range(4):
                        ok[kt, :, e] = bool(pure) | (bool(nd) & (kpix[kt][e] == qpix[qt]))
                scm = np.stack(sc)                                    # [3][64][4]
                mx = np.where(ok, scm, -1e30).max(axis=(0, 2))        # per lane
                mx = mx.reshape(4, 16).max(0)[n16]                    # across lane quads (xor 16, 32)
                pe = np.where(ok, np.exp2(scm - mx[None, :, None]), 0.0).astype(np.float32)
                rs = pe.sum(axis=(0, 2)).reshape(4, 16).sum(0)[n16]
                for kt in range(3):
                    pt[qt][kt] = h16(pe[kt] / rs[:, None])
            for m in range(KM):
                s = hs * 2 * KM + KM + m
                u0 = self.ub(s)
                for kk in range(2):
                    wv = [self.wfrag(u0, 4 * kk + j) for j in range(4)]
                    x = [self.xfrag(xn, i, 2 * m + kk) for i in range(3)]
                    for j in range(4):
                        for i in range(3):
                            av[i][j] = mfma32(x[i], wv[j], av[i][j])
            vh = [[h16(av[i][j]) for j in range(4)] for i in range(3)]
            for qt in range(3):
                for kk in range(2):
                    o0, o1 = z.copy(), z.copy()
                    for kt in range(3):
                        if (need >> (3 * qt + kt)) & 1:
                            o0 = mfma16(vh[kt][2 * kk], pt[qt][kt], o0)
                            o1 = mfma16(vh[kt][2 * kk + 1], pt[qt][kt], o1)
                    oh[hs][qt][kk] = h16(np.concatenate([o0, o1], 1))   # [64][8]
        # output projection
        out = np.zeros((48, self.inner), np.float32)
        p1s = 2 * KM * H
        ncg = self.ncgf + 1
        for c in range(ncg):
            nt = 8 if c < self.ncgf else 4
            acc = [[z.copy() for _ in range(nt)] for _ in range(3)]
            for hs in range(H):
                s = p1s + c * H + hs
                u0 = self.ub(s)
                for kk in range(2):
                    if c < self.ncgf:
                        w = [self.wfrag(u0 + kk, j) for j in range(nt)]
                    else:
                        w = [self.wfrag(u0, 4 * kk + j) for j in range(nt)]
                    for j in range(nt):
                        for i in range(3):
                            acc[i][j] = mfma32(w[j], oh[hs][i][kk], acc[i][j])
            for a in range(nt // 2):
                for i in range(3):
                    for l in range(64):
                        cb = c * 128 + 8 * q4[l] + 32 * a
                        row = 16 * i + n16[l]
                        v = np.concatenate([acc[i][2 * a][l], acc[i][2 * a + 1][l]]) + self.bo2[cb:cb + 8] + resid[row, cb:cb + 8]
                        out[row, cb:cb + 8] = h16(v)
        return out


def p0(t_rows, eps):
    """P0 of the kernel on fp16-valued rows: centre, scale, round to fp16."""
    x = np.asarray(t_rows, np.float32)
    mean = x.mean(1, keepdims=True)
    var = ((x - mean) ** 2).mean(1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    return h16(x * rstd + (-mean * rstd))
