#!/usr/bin/env python3
"""CPU interpreter for the chunk bodies tools/gen_wino4m.py emits (development / test aid).

Runs the emitted instruction text for ONE wave (64 lanes as numpy vectors) over a sequence of chunks whose LDS images are
random, and compares the accumulators with a direct evaluation of the Winograd F(4x4,3x3) chunk sum
  acc[h][p][cout][tile] += sum_k U[k][p][16 h + cout] * (B^T d B)[k][tile][p].
LDS reads are modelled with their latency semantics: a ds_read poisons its destination registers with NaN and the data
only appears once an `s_waitcnt lgkmcnt(n)` retires it (in-order return), so a missing or too weak wait shows up as NaN /
wrong sums.  LDS-DMA requests, barriers and vmcnt waits are checked for form only (the ring protocol is the kernel's).
"""
import re
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import gen_wino4m as G  # noqa: E402

BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
               [0, 4, 0, -5, 0, 1]], dtype=np.float64)


class Wave:
    def __init__(self, trow, lds_floats):
        self.v = np.zeros((256, 64), np.float32)
        self.a = np.zeros((256, 64), np.float32)
        self.lds = lds_floats                       # float32 view of the workgroup's LDS
        self.pending = []                           # (dst regs, data) oldest first
        self.ops = {}
        self.trow = trow
        self.counts = {"mfma": 0, "valu": 0, "ds": 0, "dma": 0, "barrier": 0}
        self.s = {}                                 # pinned scalar registers the text names literally
        self.scc = False

    def reg(self, tok):
        """-> (file, first, count)"""
        m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
        if m:
            return m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1
        m = re.fullmatch(r"([va])(\d+)", tok)
        assert m, tok
        return m.group(1), int(m.group(2)), 1

    def src(self, tok):
        neg = tok.startswith("-") and not re.fullmatch(r"-\d+(\.\d+)?", tok)
        if neg:
            tok = tok[1:]
        if re.fullmatch(r"-?\d+(\.\d+)?", tok):
            val = np.full(64, np.float32(float(tok)))
        elif tok.startswith("%["):
            val = self.ops[tok[2:-1]]
            val = np.full(64, val, dtype=np.float32) if np.isscalar(val) else val
        else:
            f, n, _ = self.reg(tok)
            val = (self.v if f == "v" else self.a)[n]
        return -val if neg else val

    def retire(self, keep):
        while len(self.pending) > keep:
            regs, data = self.pending.pop(0)
            for k, r in enumerate(regs):
                self.v[r] = data[k]

    def run(self, lines, operands):
        self.ops = operands
        skip_to = None
        for ln in lines:
            ln = ln.strip()
            if skip_to is not None:
                if ln == skip_to + ":":
                    skip_to = None
                continue
            if re.fullmatch(r"\d+:", ln):
                continue
            op, _, rest = ln.partition(" ")
            args = [x.strip() for x in rest.split(",")] if rest else []
            if op == "v_mfma_f32_16x16x4_f32":
                self.counts["mfma"] += 1
                f, d0, n = self.reg(args[0])
                assert n == 4 and args[3] in (args[0], "0")
                A, B = self.src(args[1]).astype(np.float64), self.src(args[2]).astype(np.float64)
                bank = self.v if f == "v" else self.a
                D = np.zeros((16, 16), np.float64)                  # [cout row][tile col]
                for r in range(4):
                    for l in range(64):
                        D[4 * (l >> 4) + r, l & 15] = 0.0 if args[3] == "0" else bank[d0 + r, l]
                for k in range(4):
                    a_col = np.array([A[16 * k + i] for i in range(16)])         # A[i][k]: lane 16 k + i
                    b_row = np.array([B[16 * k + j] for j in range(16)])         # B[k][j]: lane 16 k + j
                    D = (D + np.outer(a_col, b_row)).astype(np.float32).astype(np.float64)
                for r in range(4):
                    for l in range(64):
                        bank[d0 + r, l] = D[4 * (l >> 4) + r, l & 15]
            elif op in ("v_fma_f32", "v_add_f32", "v_sub_f32"):
                self.counts["valu"] += 1
                _, d, _ = self.reg(args[0])
                s = [self.src(x).astype(np.float64) for x in args[1:]]
                if op == "v_fma_f32":
                    res = s[0] * s[1] + s[2]
                elif op == "v_add_f32":
                    res = s[0] + s[1]
                else:
                    res = s[0] - s[1]
                self.v[d] = res.astype(np.float32)
            elif op in ("v_pk_fma_f32", "v_pk_add_f32"):
                self.counts["valu"] += 1
                nsrc = 3 if op == "v_pk_fma_f32" else 2
                body, *mods = re.split(r"\s+(?=op_sel|neg_)", rest)
                toks = [x.strip() for x in re.split(r",\s*(?![^\[]*\])", body)]
                m = {"op_sel": [0] * nsrc, "op_sel_hi": [1] * nsrc, "neg_lo": [0] * nsrc, "neg_hi": [0] * nsrc}
                for md in mods:
                    k, v = md.split(":")
                    m[k] = [int(t) for t in v.strip("[] ").split(",")]
                _, d, n = self.reg(toks[0])
                assert n == 2 and d % 2 == 0, ln

                def half(tok, sel):
                    ms = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
                    if ms:                                             # scalar pair written by the statement's own s_mov
                        return np.full(64, np.float64(np.array([self.s[int(ms.group(1)) + sel]], np.uint32).view(np.float32)[0]))
                    if re.fullmatch(r"-?\d+\.\d+", tok):             # inline constant: low half only
                        return np.full(64, np.float64(float(tok)) if sel == 0 else 0.0)
                    if tok.startswith("%["):
                        return np.full(64, np.float64(self.ops[tok[2:-1]][sel]))
                    _, r, nn = self.reg(tok)
                    assert nn == 2 and r % 2 == 0, ln
                    return self.v[r + sel].astype(np.float64)
                res = []
                for hi, (sel, neg) in enumerate(((m["op_sel"], m["neg_lo"]), (m["op_sel_hi"], m["neg_hi"]))):
                    srcs = [half(toks[1 + k], sel[k]) * (-1.0 if neg[k] else 1.0) for k in range(nsrc)]
                    res.append((srcs[0] * srcs[1] + srcs[2]) if nsrc == 3 else (srcs[0] + srcs[1]))
                self.v[d], self.v[d + 1] = res[0].astype(np.float32), res[1].astype(np.float32)
            elif op == "v_and_b32":
                _, d, _ = self.reg(args[0])
                self.v[d] = (self.src_int(args[1]) & self.src_int(args[2])).astype(np.uint32).view(np.float32)
            elif op == "v_cmp_ne_u32":
                assert args[0] == "vcc"
                self.vcc = self.src_int(args[1]) != self.src_int(args[2])
            elif op == "v_cndmask_b32":
                assert args[3] == "vcc"
                _, d, _ = self.reg(args[0])
                self.v[d] = np.where(self.vcc, self.src(args[2]), self.src(args[1])).astype(np.float32)
            elif op == "v_add_u32":
                _, d, _ = self.reg(args[0])
                a, b = self.src_int(args[1]), self.src_int(args[2])
                self.v[d] = (a + b).astype(np.uint32).view(np.float32)
            elif op in ("ds_read_b128", "ds_read_b64"):
                self.counts["ds"] += 1
                _, d, n = self.reg(args[0])
                addr_tok, off_tok = args[1].split()
                _, ar, _ = self.reg(addr_tok)
                off = int(off_tok.split(":")[1])
                addr = self.v[ar].view(np.uint32).astype(np.int64) + off
                assert n == (4 if op.endswith("128") else 2) and (addr % (4 * n if n == 4 else 8) == 0).all(), (ln, addr[:4])
                data = [self.lds[(addr // 4 + k)] .copy() for k in range(n)]
                for k in range(n):
                    self.v[d + k] = np.nan
                self.pending.append((list(range(d, d + n)), data))
            elif op == "s_waitcnt":
                m = re.fullmatch(r"lgkmcnt\((\d+)\)", args[0])
                if m:
                    self.retire(int(m.group(1)))
                else:
                    assert re.fullmatch(r"vmcnt\(\d+\)", args[0]), ln
            elif op == "s_barrier":
                self.counts["barrier"] += 1
            elif op == "buffer_load_dwordx4":
                self.counts["dma"] += 1
                assert rest.endswith("offen lds"), ln
            elif op == "s_add_u32" and args[0].startswith("%["):
                pass                                             # (weight piece offsets: not modelled)
            elif op in ("s_add_u32", "s_sub_u32", "s_addc_u32") and re.fullmatch(r"s\d+", args[0]):
                a, b = self.sval(args[1]), self.sval(args[2])
                r = a + b + (1 if op == "s_addc_u32" and self.scc else 0) if op != "s_sub_u32" else a - b
                self.scc = r >= (1 << 32) or r < 0
                self.s[int(args[0][1:])] = r & 0xffffffff
            elif op == "s_mov_b32" and re.fullmatch(r"s\d+", args[0]):
                self.s[int(args[0][1:])] = self.sval(args[1])
            elif op == "s_cselect_b32":
                self.s[int(args[0][1:])] = self.sval(args[1]) if self.scc else self.sval(args[2])
            elif op == "s_cmp_ge_u32":
                self.scc = self.sval(args[0]) >= self.sval(args[1])
            elif op == "s_cmp_lt_u32":
                self.scc = self.sval(args[0]) < self.sval(args[1])
            elif op == "s_bitcmp1_b32":
                self.scc = bool((int(self.ops[args[0][2:-1]]) >> int(args[1])) & 1)
            elif op == "s_cbranch_scc0":
                if not self.scc:
                    skip_to = args[0][:-1]
            elif op == "s_cbranch_scc1":
                if self.scc:
                    skip_to = args[0][:-1]
            elif op == "s_branch":
                skip_to = args[0][:-1]
            elif op in ("s_add_i32", "s_nop"):
                pass
            else:
                raise AssertionError("unknown instruction: " + ln)

    def sval(self, tok):
        if re.fullmatch(r"s\d+", tok):
            return int(self.s.get(int(tok[1:]), 0))
        if tok.startswith("%["):
            return int(self.ops[tok[2:-1]])
        return int(tok, 0)

    def src_int(self, tok):
        if re.fullmatch(r"s\d+", tok):
            return np.full(64, self.sval(tok), dtype=np.int64)
        if re.fullmatch(r"\d+", tok):
            return np.full(64, int(tok), dtype=np.int64)
        if tok.startswith("%["):
            val = self.ops[tok[2:-1]]
            return np.full(64, val, dtype=np.int64) if np.isscalar(val) else val.view(np.uint32).astype(np.int64)
        _, n, _ = self.reg(tok)
        return self.v[n].view(np.uint32).astype(np.int64)


def reference_chunk(lds, slot, trow, mode=0, fixmask=None):
    """-> contribution [h][p][cout 16][tile 16] of the chunk in ring slot `slot` for tile row trow (mode 1 / 2: the
    border fix-up with zero / reflect padding applied to the columns in each lane's fixmask)."""
    base = slot * (G.IN_FLOATS + G.W_FLOATS)
    out = np.zeros((2, 36, 16, 16), np.float64)
    V = np.zeros((4, 16, 36), np.float64)
    for k in range(4):
        for n in range(16):
            o = base + k * G.PLANE_S + 4 * trow * G.ROWP + 4 * n
            d = np.array([[lds[o + r * G.ROWP + j] for j in range(6)] for r in range(6)], np.float64)
            if mode:
                for i in range(6):
                    if (int(fixmask[16 * k + n]) >> (8 + i)) & 1:
                        d[i, 1:4] = d[i, 0:3].copy()
                raw = d.copy()
                for j in range(6):
                    if (int(fixmask[16 * k + n]) >> j) & 1:
                        d[:, j] = 0.0 if mode == 1 else raw[:, 2 if j == 0 else j - 2]
            V[k, n] = (BT @ d @ BT.T).reshape(36)
    for h in range(2):
        for p in range(36):
            for k in range(4):
                U = np.array([lds[base + G.IN_FLOATS + ((k * G.NGRP + p // 4) * G.BN + 16 * h + c) * 4 + p % 4] for c in range(16)], np.float64)
                out[h, p] += np.outer(U, V[k, :, p])
    return out


def simulate(nchunks=5, trow=2, wave=1, seed=0, mode=0):
    rng = np.random.default_rng(seed)
    prime, b0, b1, b0f, b1f = G.main("/dev/null")
    lds = np.zeros(G.NBUF * (G.IN_FLOATS + G.W_FLOATS) + 512, np.float32)
    wv = Wave(trow, lds)
    wv.a[:] = 123.0                                  # (the first chunk's statements must not read the accumulators)
    wv.v[G.ACCV:] = 321.0
    lane = np.arange(64)
    n16, k4 = lane & 15, lane >> 4
    pa0 = ((k4 * G.PLANE_S + 4 * trow * G.ROWP + 4 * n16) * 4).astype(np.uint32).view(np.float32)
    wa0 = ((G.IN_FLOATS + (k4 * G.NGRP * G.BN + n16) * 4) * 4).astype(np.uint32).view(np.float32)
    want = np.zeros((2, 36, 16, 16), np.float64)
    if mode == 1:
        fixmask = rng.integers(0, 64, 64).astype(np.uint32)
    else:                                                           # reflect: column -1 (bit 0) or the one column W (bit 2..5)
        fixmask = np.where(n16 == 0, 1, np.where(n16 == 15, 1 << rng.integers(2, 6, 64), 0)).astype(np.uint32)
    if mode:
        fixmask[0] |= 1 | (256 << 1)                                # lane (n16 = 0, k4 = 0): row 1 fetched one column off
        fixmask[5] |= 256 << 4
    rflags = 0 if mode == 0 else (2 | (4 if mode == 2 else 0))

    def operands(slot_rd):
        assert wv.s.get(G.S_RD, 0) == slot_rd * G.BUF_BYTES, (wv.s, slot_rd)       # the read cursor is the statements' own
        return {"pa0": pa0, "wa0": wa0, "s_five": 5.0, "s_wave": wave, "s_rflags": rflags, "fixmask": fixmask.view(np.float32),
                "k4": (4.0, 4.0), "k5": (5.0, 5.0), "k2": (2.0, 2.0), "k41": (4.0, 1.0), "s_inc": 4 * 1920 * 1080 * 4, "s_winc": 4 * 36 * 4 * 64, "s_dma_end": 4 * G.BUF_BYTES + wave * 1024}

    def fill(slot):
        b = slot * (G.IN_FLOATS + G.W_FLOATS)
        lds[b:b + G.IN_FLOATS + G.W_FLOATS] = rng.standard_normal(G.IN_FLOATS + G.W_FLOATS).astype(np.float32)

    wv.s[G.S_DMA] = wave * 1024
    fill(0)
    chunk_slots = [0]
    wv.run(prime, operands(0))
    for c in range(nchunks):
        want += reference_chunk(lds, chunk_slots[c], trow, mode, fixmask)
        nxt = (c + 1) % G.NBUF
        fill(nxt)                                  # chunk c+1 has landed by the time body c reads it
        chunk_slots.append(nxt)
        wv.run((b0f if c == 0 else b0) if c % 2 == 0 else b1, operands(nxt))
    wv.retire(0)
    got = np.zeros_like(want)
    for h in range(2):
        for p in range(36):
            k = 36 * h + p
            bank, base = (wv.a, 4 * k) if k < 64 else (wv.v, G.ACCV + 4 * (k - 64))
            for r in range(4):
                for l in range(64):
                    got[h, p, 4 * (l >> 4) + r, l & 15] = bank[base + r, l]
    err = np.abs(got - want).max() / np.abs(want).max()
    return err, wv.counts


if __name__ == "__main__":
    for mode in (0, 1, 2):
        err, counts = simulate(mode=mode, nchunks=3 + mode, trow=mode + 1, wave=mode)
        print("mode", mode, "relative error", err, counts)
        assert err < 1e-5
