#!/usr/bin/env python
"""CPU check of the index arithmetic of tools/gemm_bf16x6.hip (written without a GPU at hand): the LDS-DMA image with its
global-side XOR swizzles, the per-lane fragment offsets, the 16x16x32 MFMA operand / result lane maps and the store addresses are
re-stated here with the SAME integer expressions and run on a small problem in numpy.  It proves the pieces are consistent with
each other and with out = x . w^T (it cannot prove the hardware lane maps, which are taken from the CDNA4 guide; the GPU harness'
identity check does that).     python tools/gemm_bf16x6_indexcheck.py"""
import numpy as np

BK, NW = 32, 8


def main(WM, WN, RB, ntok, scenes_total, tail=0, pipe3=False):
    global BM, BN, A_STAGE, B_PLANE
    FR.clear(); FW.clear()
    BM, BN = 16 * RB * WM, 64 * WN                          # LDS rows (scenes padded to 16*RB); global rows per block = WM * ntok
    A_STAGE, B_PLANE = BM * BK * 4, BN * BK * 2
    X_PLANE = BM * BK * 2
    XA = 3 * X_PLANE if pipe3 else A_STAGE                  # PIPE 3: the x region holds three bf16 planes written by the staging threads
    CH_A, CH_PL = (0 if pipe3 else A_STAGE // 1024), B_PLANE // 1024
    CH = CH_A + 3 * CH_PL
    NI = (CH + NW - 1) // NW
    rng = np.random.default_rng(0)
    M, N, K, K1 = scenes_total * ntok - tail, 2 * BN, 96, 64   # two K segments: 64 columns from x, 32 from x2; tail: dense GEMM whose m is not a multiple of 16*RB
    lda = 64 + 8                                            # a row stride that is not K (the same for both segments)
    x = rng.standard_normal((M, lda)).astype(np.float32)
    x2 = rng.standard_normal((M, lda)).astype(np.float32)
    x2bytes = x2.view(np.uint8).reshape(-1)
    # "planes": any three distinct matrices stand in for the bf16 pieces (kept as f32 values here; 2-byte elements in the byte image)
    planes = rng.integers(-50, 50, size=(3, N, K)).astype(np.int16)
    xbytes = x.view(np.uint8).reshape(-1)
    pbytes = planes.view(np.uint8).reshape(-1)
    out = np.zeros((M, N), np.float64)
    written = np.zeros((M, N), bool)
    rbs, cbs = (scenes_total + WM - 1) // WM, N // BN
    for blk in range(rbs * cbs):
        rb, cb = blk // cbs, blk % cbs                      # (the XCD remap only permutes blocks)
        row0, col0 = rb * WM * ntok, cb * BN
        rows_here = M - row0
        xb = row0 * lda * 4                                 # byte offset of the block's first token row
        wb = col0 * K * 2                                   # byte offset of the block's first channel row in plane 0
        plane_bytes = N * K * 2
        acc = np.zeros((NW, RB, 4, 64, 4), np.float64)
        for kt in range(K // BK):
            k0 = kt * BK
            lds = np.zeros(XA + 3 * B_PLANE, np.uint8)
            xpl = {}                                        # PIPE 3: (plane, byte offset) -> 8 values (exact stand-ins for the bf16 pieces)
            for wave in range(NW):
                for i in range(NI):
                    c = wave + NW * i
                    if c >= CH:
                        c -= CH
                    for lane in range(64):
                        if c < CH_A:
                            r = c * 8 + (lane >> 3)
                            sc, tk = r // (16 * RB), r % (16 * RB)
                            gr = sc * ntok + tk
                            voff = (gr if (tk < ntok and gr < rows_here) else 0) * lda * 4 + (((lane & 7) ^ (r & 7)) << 4)
                            seg1 = k0 < K1
                            sb, sx = (xbytes, k0 * 4) if seg1 else (x2bytes, (k0 - K1) * 4)
                            src = sb[xb + voff + sx: xb + voff + sx + 16]
                        else:
                            cbk = c - CH_A
                            plane, nrow = cbk // CH_PL, (cbk % CH_PL) * 16 + (lane >> 2)
                            voff = plane * plane_bytes + nrow * K * 2 + (((lane & 3) ^ ((nrow >> 1) & 3)) << 4)
                            src = pbytes[wb + voff + k0 * 2: wb + voff + k0 * 2 + 16]
                        dst = c * 1024 if c < CH_A else XA + (c - CH_A) * 1024
                        lds[dst + lane * 16: dst + lane * 16 + 16] = src
                if pipe3:
                    ITEMS_W = BM * 4 // NW
                    for u in range((ITEMS_W + 63) // 64):
                        for lane in range(64):
                            idx = 64 * u + lane
                            if idx >= ITEMS_W:
                                continue                    # (the kernel sends these lanes' copies of item 0 to a dump slot)
                            t = wave * ITEMS_W + idx
                            r, q = t >> 2, t & 3
                            sc, tk = r // (16 * RB), r % (16 * RB)
                            gr = sc * ntok + tk
                            ivoff = (gr if (tk < ntok and gr < rows_here) else 0) * lda * 4 + q * 32
                            seg1 = k0 < K1
                            sb, sx = (xbytes, k0 * 4) if seg1 else (x2bytes, (k0 - K1) * 4)
                            v = sb[xb + ivoff + sx: xb + ivoff + sx + 32].view(np.float32).astype(np.float64)
                            ildso = r * 64 + ((q ^ ((r >> 1) & 3)) << 4)
                            for pl, part in enumerate((np.round(v * 4) / 4, v - np.round(v * 4) / 4, np.zeros(8))):
                                assert (pl, ildso) not in xpl
                                xpl[pl, ildso] = part
            for wave in range(NW):
                wm, wn = wave % WM, wave // WM
                for lane in range(64):
                    g, l15 = lane >> 4, lane & 15
                    wf = np.zeros((4, 3, 8), np.float64)
                    for j in range(4):
                        nr = wn * 64 + j * 16 + l15
                        woff = XA + nr * 64 + ((g ^ ((nr >> 1) & 3)) << 4)
                        for pl in range(3):
                            wf[j, pl] = lds[woff + pl * B_PLANE: woff + pl * B_PLANE + 16].view(np.int16)
                    acc_lane_w = wf
                    for i in range(RB):
                        r = (wm * RB + i) * 16 + l15
                        a0 = r * 128 + (((2 * g) ^ (r & 7)) << 4)
                        a1 = r * 128 + (((2 * g + 1) ^ (r & 7)) << 4)
                        if pipe3:
                            xo = r * 64 + ((g ^ ((r >> 1) & 3)) << 4)
                            xf = xpl[0, xo] + xpl[1, xo] + xpl[2, xo]
                        else:
                            xf = np.concatenate([lds[a0:a0 + 16].view(np.float32), lds[a1:a1 + 16].view(np.float32)]).astype(np.float64)
                        # stash the fragments; the MFMA is evaluated below over all lanes of the wave
                        FR.setdefault((wave, i), np.zeros((64, 8)))[lane] = xf
                    FW[wave, lane] = acc_lane_w
            # MFMA 16x16x32: D[row][col] += sum_k A[row][k] B[k][col]; lane l holds A[l&15][8*(l>>4)+e], B[8*(l>>4)+e][l&15];
            # result lane l, reg e: row 4*(l>>4)+e, col l&15.  A = weight fragment (plane sum stands for the six products), B = tokens.
            for wave in range(NW):
                for i in range(RB):
                    for j in range(4):
                        Amat = np.zeros((16, 32))
                        Bmat = np.zeros((32, 16))
                        for lane in range(64):
                            g, l15 = lane >> 4, lane & 15
                            Amat[l15, 8 * g: 8 * g + 8] = FW[wave, lane][j].sum(axis=0)     # w1 + w2 + w3
                            Bmat[8 * g: 8 * g + 8, l15] = FR[(wave, i)][lane]
                        D = Amat @ Bmat
                        for lane in range(64):
                            g, l15 = lane >> 4, lane & 15
                            acc[wave, i, j, lane] += D[4 * g: 4 * g + 4, l15]
        for wave in range(NW):
            wm, wn = wave % WM, wave // WM
            for lane in range(64):
                g, l15 = lane >> 4, lane & 15
                srow = wm * ntok
                for i in range(RB):
                    if not (i * 16 + l15 < ntok and srow + i * 16 + l15 < rows_here):
                        continue
                    for j in range(4):
                        n = col0 + wn * 64 + 4 * g + j * 16
                        assert not written[row0 + srow + i * 16 + l15, n]
                        written[row0 + srow + i * 16 + l15, n: n + 4] = True
                        out[row0 + srow + i * 16 + l15, n: n + 4] = acc[wave, i, j, lane]
    xcat = np.concatenate([x[:, :K1], x2[:, :K - K1]], axis=1)
    ref = xcat.astype(np.float64) @ planes.astype(np.float64).sum(axis=0).T
    assert written.all(), "some outputs were never stored"
    err = np.abs(out - ref).max()
    print("%swaves %d x %d, RB %d, %d tokens/scene, %d scenes: max |emulated kernel - [x | x2].w^T| = %.3e over %d x %d outputs (K = %d + %d)" % (
        "PIPE 3 (staged planes) " if pipe3 else "", WM, WN, RB, ntok, scenes_total, err, M, N, K1, K - K1))
    assert err < 1e-9 * max(1.0, np.abs(ref).max())
    print("index arithmetic consistent")


def bank_check(WM, WN, RB):
    """ds_read_b128: the LDS serves 128 B per clock = 8 lanes x 16 B; conflict-free when each run of 8 consecutive lanes touches
    8 distinct 16-byte bank groups ((address / 16) mod 8)."""
    worst = 1
    A_STAGE = 16 * RB * WM * BK * 4
    for wave in range(NW):
        wm, wn = wave % WM, wave // WM
        for i in range(RB):
            for half in range(2):
                ad = []
                for lane in range(64):
                    g, l15 = lane >> 4, lane & 15
                    r = (wm * RB + i) * 16 + l15
                    ad.append(r * 128 + (((2 * g + half) ^ (r & 7)) << 4))
                for q in range(8):
                    grp = [(a >> 4) & 7 for a in ad[8 * q: 8 * q + 8]]
                    worst = max(worst, max(grp.count(v) for v in grp))
        for j in range(4):
            ad = []
            for lane in range(64):
                g, l15 = lane >> 4, lane & 15
                nr = wn * 64 + j * 16 + l15
                ad.append(A_STAGE + nr * 64 + ((g ^ ((nr >> 1) & 3)) << 4))
            for q in range(8):
                grp = [(a >> 4) & 7 for a in ad[8 * q: 8 * q + 8]]
                worst = max(worst, max(grp.count(v) for v in grp))
    print("waves %d x %d, RB %d fragment reads: worst bank-group multiplicity within 8 consecutive lanes = %d (1 = conflict-free)" % (WM, WN, RB, worst))
    assert worst == 1


def tn_check():
    """tools/gemm_tn_bf16x6.hip: staging items -> channel-major planes -> fragments -> MFMA (x as row operand) -> stores."""
    BN_, BKO, COLS, T = 256, 128, 384, 512
    rng = np.random.default_rng(1)
    M, N, K, slices = 128, 256, 256, 2                      # 2 steps per slice; grid 1 x 2 x 2
    ldy, ldx = N + 4, K + 12
    dy = rng.integers(-8, 8, size=(M, ldy)).astype(np.float64)
    x = rng.integers(-8, 8, size=(M, ldx)).astype(np.float64)
    rows_per_slice = M // slices
    out = np.zeros((slices, N, K))
    written = np.zeros((slices, N, K), bool)
    for bx in range(N // BN_):
        for by in range(K // BKO):
            for sl in range(slices):
                n0, k0, m_begin = bx * BN_, by * BKO, sl * rows_per_slice
                acc = np.zeros((8, 4, 4, 64, 4))
                for step in range(rows_per_slice // 32):
                    mb = m_begin + step * 32
                    planes = {}
                    for u in range(3):
                        for tid in range(T):
                            wave = tid >> 6
                            t = u * T + tid
                            c, og = t % COLS, t // COLS
                            isdy = ((u * T + wave * 64) % COLS) < BN_
                            assert isdy == (c < BN_)        # the wave-uniform test agrees with every lane of the wave
                            if isdy:
                                v = np.array([dy[mb + 8 * og + e, n0 + c] for e in range(8)])
                            else:
                                v = np.array([x[mb + 8 * og + e, k0 + c - BN_] for e in range(8)])
                            ildso = c * 64 + ((og ^ ((c >> 1) & 3)) << 4)
                            assert ildso not in planes
                            planes[ildso] = v
                    for wave in range(8):
                        wn, wk = wave & 3, wave >> 2
                        xf = np.zeros((4, 64, 8)); df = np.zeros((4, 64, 8))
                        for lane in range(64):
                            g, l15 = lane >> 4, lane & 15
                            for b in range(4):
                                cx, cd = BN_ + wk * 64 + b * 16 + l15, wn * 64 + b * 16 + l15
                                xf[b, lane] = planes[cx * 64 + ((g ^ ((cx >> 1) & 3)) << 4)]
                                df[b, lane] = planes[cd * 64 + ((g ^ ((cd >> 1) & 3)) << 4)]
                        for kb in range(4):
                            for nb in range(4):
                                A = np.zeros((16, 32)); B = np.zeros((32, 16))
                                for lane in range(64):
                                    g, l15 = lane >> 4, lane & 15
                                    A[l15, 8 * g: 8 * g + 8] = xf[kb, lane]          # row operand: x channel (k)
                                    B[8 * g: 8 * g + 8, l15] = df[nb, lane]          # column operand: dy channel (n)
                                D = A @ B
                                for lane in range(64):
                                    g, l15 = lane >> 4, lane & 15
                                    acc[wave, kb, nb, lane] += D[4 * g: 4 * g + 4, l15]
                for wave in range(8):
                    wn, wk = wave & 3, wave >> 2
                    for lane in range(64):
                        g, l15 = lane >> 4, lane & 15
                        for nb in range(4):
                            for kb in range(4):
                                n = n0 + wn * 64 + l15 + nb * 16
                                k = k0 + wk * 64 + 4 * g + kb * 16
                                assert not written[sl, n, k]
                                written[sl, n, k: k + 4] = True
                                out[sl, n, k: k + 4] = acc[wave, kb, nb, lane]
    ref = dy[:, :N].T @ x[:, :K]
    err = np.abs(out.sum(axis=0) - ref).max()
    assert written.all()
    print("TN: max |emulated kernel - dy^T.x| = %.3e over %d x %d outputs (%d tokens in %d slices)" % (err, N, K, M, slices))
    assert err == 0


FR, FW = {}, {}
if __name__ == "__main__":
    tn_check()
    # (WM, WN, RB, tokens per scene, scenes): the 80-token tiles, a ragged scene count, 21-token scenes, a dense GEMM with a row tail
    for cfg in ((2, 4, 5, 80, 4), (4, 2, 5, 80, 5), (4, 2, 2, 21, 7), (2, 4, 5, 80, 3), (2, 4, 5, 80, 4, 70)):
        main(*cfg)
        bank_check(*cfg[:3])
    for cfg in ((2, 4, 5, 80, 4), (4, 2, 2, 21, 7), (2, 4, 5, 80, 4, 70)):
        main(*cfg, **({"pipe3": True} if len(cfg) == 6 else {"tail": 0, "pipe3": True}))
