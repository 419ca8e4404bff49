"""Test infrastructure: a lane-level NumPy walk through csrc/xattn.hip (K5) for ONE wave — the kernel's own unit / tile / slot
arithmetic for the q stream and the output projection (shared with tattn2.hip: tests/k7b_emulator.py), its addressing of the
per-prompt key / value fragment blobs, the MFMA lane maps and the accumulator-as-operand hand-offs — on the blobs
`packing.pack_k5` / `packing.pack_k5_kv` produce, so that host packing and device indexing are checked against each other
without a GPU.  Not product code."""
from __future__ import annotations

import numpy as np

from k7b_emulator import UB, h16, mfma16, mfma32

NU, KM, H, KVU, KT = 5, 5, 5, 3, 5
UPH, HSTEPS = KM + KVU, KM + 1
P1S = H * HSTEPS


class Wave5:
    def __init__(self, blob_f16: np.ndarray, kv_f16: np.ndarray, item: int, rot: int, kv_len: int):
        raw = blob_f16.view(np.uint8)
        nstatic = KM * H + 2 * H * 2 + H                      # 25 q units + 25 output-projection units
        self.wq = raw[: KM * H * UB]
        self.wo = raw[KM * H * UB: nstatic * UB]
        vec = raw[nstatic * UB:].view(np.float32)
        self.bq, self.bo = vec[:320], vec[320:640]
        self.kv = kv_f16.view(np.uint8).reshape(-1)           # [item][head][3 units]
        self.item, self.rot, self.kv_len = item, rot, kv_len
        lane = np.arange(64)
        self.n16, self.q4 = lane & 15, lane >> 4
        g = np.array([0, 2, 3, 1])[self.n16 >> 2]
        self.woff = self.n16 * 64 + ((self.q4 ^ g) << 4)

    def ub(self, s):
        if s <= P1S:
            return UPH * (s // HSTEPS) + (s % HSTEPS)
        v = s - P1S
        c, m = divmod(v, H)
        return UPH * H + (2 * H * c + 2 * m if c < 2 else 4 * H + (v - 2 * H))

    def unit_bytes(self, u):
        """unit_src<U>() of the kernel."""
        if u < UPH * H:
            hs, w = divmod(u, UPH)
            h = (hs + self.rot) % H
            if w < KM:
                off = (h * KM + w) * UB
                return self.wq[off:off + UB]
            off = ((self.item * H + h) * KVU + (w - KM)) * UB
            return self.kv[off:off + UB]
        v = u - UPH * H
        if v < 4 * H:
            cg, r = divmod(v, 2 * H)
            hs, kk = divmod(r, 2)
            off = (cg * 2 * H + ((hs + self.rot) % H) * 2 + kk) * UB
        else:
            off = (4 * H + ((v - 4 * H + self.rot) % H)) * UB
        return self.wo[off:off + UB]

    def wfrag(self, u, tile):
        unit = self.unit_bytes(u)
        out = np.zeros((64, 8), np.float32)
        for l in range(64):
            o = tile * 1024 + self.woff[l]
            out[l] = unit[o:o + 16].view(np.float16).astype(np.float32)
        return out

    def kvfrag(self, u0, blk):
        """kvfrag<U0, BLK>(): byte BLK * 512 of the head's 3-unit stream, 8 bytes per lane."""
        byte = blk * 512
        unit = self.unit_bytes(u0 + byte // UB)
        out = np.zeros((64, 4), np.float32)
        for l in range(64):
            o = byte % UB + l * 8
            out[l] = unit[o:o + 8].view(np.float16).astype(np.float32)
        return out

    @staticmethod
    def xfrag(xn, i, ks, n16, q4):
        out = np.zeros((64, 8), np.float32)
        for l in range(64):
            c0 = 8 * (4 * ks + q4[l])
            out[l] = xn[16 * i + n16[l], c0:c0 + 8]
        return out

    def run(self, xn, resid):
        """xn: the wave's 48 rows after P0 (fp16 values), resid: the raw rows -> [48][320] output rows (fp16 values)."""
        n16, q4 = self.n16, self.q4
        z = np.zeros((64, 4), np.float32)
        oh = [[[None, None] for _ in range(3)] for _ in range(H)]
        for hs in range(H):
            h = (hs + self.rot) % H
            bqv = [np.stack([self.bq[h * 64 + 16 * j + 4 * q4 + e] for e in range(4)], 1) for j in range(4)]
            aq = [[bqv[j].copy() for j in range(4)] for _ in range(3)]
            for m in range(KM):
                u0 = self.ub(hs * HSTEPS + m)
                for kk in range(2):
                    w = [self.wfrag(u0, 4 * kk + j) for j in range(4)]
                    x = [self.xfrag(xn, i, 2 * m + kk, n16, q4) for i in range(3)]
                    for j in range(4):
                        for i in range(3):
                            aq[i][j] = mfma32(w[j], x[i], aq[i][j])
            u0 = self.ub(hs * HSTEPS + KM)
            qh = [[h16(aq[i][j]) for j in range(4)] for i in range(3)]
            kf = [[self.kvfrag(u0, kt * 4 + j) for j in range(4)] for kt in range(KT)]
            vh = [[self.kvfrag(u0, KT * 4 + kt * 4 + dt) for dt in range(4)] for kt in range(KT)]
            kl = self.kv_len - 4 * q4
            for qt in range(3):
                sc = []
                for kt in range(KT):
                    c = z.copy()
                    for j in range(4):
                        c = mfma16(kf[kt][j], qh[qt][j], c)
                    sc.append(c)
                scm = np.stack(sc)                                                    # [kt][lane][e]
                ok = (16 * np.arange(KT)[:, None, None] + np.arange(4)[None, None, :]) < kl[None, :, None]
                mx = np.where(ok, scm, -1e30).max(axis=(0, 2)).reshape(4, 16).max(0)[n16]
                pe = np.where(ok, np.exp2(scm - mx[None, :, None]), 0.0).astype(np.float32)
                rs = pe.sum(axis=(0, 2)).reshape(4, 16).sum(0)[n16]
                pt = [h16(pe[kt] / rs[:, None]) for kt in range(KT)]
                for kk in range(2):
                    o0, o1 = z.copy(), z.copy()
                    for kt in range(KT):
                        o0 = mfma16(vh[kt][2 * kk], pt[kt], o0)
                        o1 = mfma16(vh[kt][2 * kk + 1], pt[kt], o1)
                    oh[hs][qt][kk] = h16(np.concatenate([o0, o1], 1))
        out = np.zeros((48, 320), np.float32)
        for c in range(3):
            nt = 8 if c < 2 else 4
            acc = [[z.copy() for _ in range(nt)] for _ in range(3)]
            for hs in range(H):
                u0 = self.ub(P1S + c * H + hs)
                for kk in range(2):
                    w = [self.wfrag(u0 + kk, j) for j in range(nt)] if c < 2 else [self.wfrag(u0, 4 * kk + j) for j in range(nt)]
                    for j in range(nt):
                        for i in range(3):
                            acc[i][j] = mfma32(w[j], oh[hs][i][kk], acc[i][j])
            for a in range(nt // 2):
                for i in range(3):
                    for l in range(64):
                        cb = c * 128 + 8 * q4[l] + 32 * a
                        row = 16 * i + n16[l]
                        v = h16(np.concatenate([acc[i][2 * a][l], acc[i][2 * a + 1][l]]) + self.bo[cb:cb + 8])
                        out[row, cb:cb + 8] = h16(v + resid[row, cb:cb + 8])
        return out
