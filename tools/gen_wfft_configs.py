#!/usr/bin/env python3
"""Generator of csrc/vfi_wfft_configs.h: the compile-time geometry of the wave-private FFT engine (csrc/vfi_wfft.h).

For every engine length M and mode (rows: one wave owns L whole rows; cols: one wave owns L adjacent columns) it picks
  * the radix sequence (radices 2..16, as few stages as possible, then as few registers per lane as possible, first radix ==
    last radix where that is free: Bluestein's second transform then starts from the registers the first one ended in),
  * the padding of the LDS exchange buffer (one pad word per block of the stage's output stride where that removes bank
    conflicts) and, in column mode, the line pitch residue modulo the 32 banks,
by simulating the engine's own index arithmetic: every candidate is run lane by lane in numpy against numpy.fft (so a
wrong formula cannot reach the header) and its LDS accesses are priced with the bank model of MI355X_MICROARCH.md
(ds_read_b32 / ds_write_b32: two groups of 32 lanes, bank = dword address mod 32, an n-way conflict costs n cycles).

    python3 tools/gen_wfft_configs.py > fusion-method-for-video-frame-interpolation_amd/csrc/vfi_wfft_configs.h
"""
import itertools
import sys

import numpy as np

RADICES = (16, 15, 12, 10, 9, 8, 6, 5, 4, 3, 2)
WAVE = 64


def sequences(m, max_stages=4):
    out = []

    def rec(rem, seq):
        if rem == 1:
            if seq:
                out.append(tuple(seq))
            return
        if len(seq) == max_stages:
            return
        for r in RADICES:
            if rem % r == 0:
                rec(rem // r, seq + [r])
    rec(m, [])
    return out


class Geometry:
    def __init__(self, m, lines, cols, radices, team=WAVE):
        self.m, self.lines, self.cols, self.r, self.team = m, lines, cols, tuple(radices), team
        self.ns = len(radices)
        self.p = [int(np.prod(radices[:s])) for s in range(self.ns)]
        self.t = [m // r for r in radices]
        self.nb = [lines * t for t in self.t]
        self.q = [(nb + team - 1) // team for nb in self.nb]
        self.e = max(q * r for q, r in zip(self.q, self.r))
        self.work = sum(q * r for q, r in zip(self.q, self.r))

    def lane_index(self, s, lane, q):
        """-> (line, butterfly index i, valid)"""
        idx = lane + self.team * q
        if self.cols:
            return idx % self.lines, idx // self.lines, idx < self.nb[s]
        return idx // self.t[s], idx % self.t[s], idx < self.nb[s]


def choose_radices(m, lines, cols, want_symmetric, team=WAVE):
    best = None
    for seq in sequences(m):
        g = Geometry(m, lines, cols, seq, team)
        sym = 0 if (not want_symmetric or seq[0] == seq[-1]) else 1
        # a stage-0 radix that is odd needs no pad word; prefer fewer stages, then fewer registers, then less padded work
        key = (g.ns, g.e, sym, g.work, tuple(-r for r in seq))
        if best is None or key < best[0]:
            best = (key, seq)
    return best[1]


def phys(pos, block, pad):
    return pos + (pos // block) * pad


def exchange_addresses(g, s, pads, pitch):
    """Addresses (dwords) of every write instruction of stage s and every read instruction of stage s+1:
    lists of (lane -> address or None)."""
    r, p = g.r[s], g.p[s]
    block = p * r
    writes, reads = [], []
    for q in range(g.q[s]):
        for rr in range(r):
            a = []
            for lane in range(g.team):
                l, i, ok = g.lane_index(s, lane, q)
                a.append(l * pitch + phys((i // p) * block + (i % p) + rr * p, block, pads[s]) if ok else None)
            writes.append(a)
    r2, t2 = g.r[s + 1], g.t[s + 1]
    for q in range(g.q[s + 1]):
        for rr in range(r2):
            a = []
            for lane in range(g.team):
                l, i, ok = g.lane_index(s + 1, lane, q)
                a.append(l * pitch + phys(i + rr * t2, block, pads[s]) if ok else None)
            reads.append(a)
    return writes, reads


def conflict_cycles(instrs):
    cyc = 0
    for a in instrs:
        for half in [a[k:k + 32] for k in range(0, len(a), 32)]:
            banks = {}
            for x in half:
                if x is not None:
                    banks.setdefault(x % 32, set()).add(x)
            cyc += max([len(v) for v in banks.values()] or [0])
    return cyc


def line_length(g, pads):
    n = g.m
    for s in range(g.ns - 1):
        n = max(n, phys(g.m - 1, g.p[s] * g.r[s], pads[s]) + 1)
    return n


def choose_layout(g):
    """-> (pads per exchange, pitch): minimal simulated LDS cycles, then minimal buffer."""
    pads = [0] * max(g.ns - 1, 0)
    residues = (8, 24) if g.cols else (0,)
    best = None
    for res in residues:
        cur = []
        total = 0
        for s in range(g.ns - 1):
            cand = None
            for pad in range(0, 5):
                trial = cur + [pad] + [0] * (g.ns - 2 - s)
                # pitch for this trial (only its residue matters for conflicts)
                ll = line_length(g, trial)
                pitch = ((ll + 31) // 32) * 32 + res if g.cols else ll
                w, r = exchange_addresses(g, s, trial, pitch)
                cyc = conflict_cycles(w) * 2 + conflict_cycles(r)      # a write cycle costs two (microarch guide)
                key = (cyc + 40 * pad * g.lines * (g.m // (g.p[s] * g.r[s])) // 64, pad)
                if cand is None or key < cand[0]:
                    cand = (key, pad, cyc)
            cur.append(cand[1])
            total += cand[2]
        ll = line_length(g, cur) if cur else g.m
        pitch = ((ll + 31) // 32) * 32 + res if g.cols else ll
        key = (total, pitch)
        if best is None or key < best[0]:
            best = (key, cur, pitch, total)
    return best[1], best[2], best[3]


def simulate(g, pads, pitch, rng):
    """The engine's data flow in numpy (complex128): returns max |error| against numpy.fft."""
    x = rng.standard_normal((g.lines, g.m)) + 1j * rng.standard_normal((g.lines, g.m))
    v = {}
    for lane in range(g.team):
        for q in range(g.q[0]):
            l, i, ok = g.lane_index(0, lane, q)
            for rr in range(g.r[0]):
                v[(lane, q, rr)] = x[l, i + rr * g.t[0]] if ok else 0.0
    for s in range(g.ns):
        r, p, t = g.r[s], g.p[s], g.t[s]
        out = {}
        for lane in range(g.team):
            for q in range(g.q[s]):
                l, i, ok = g.lane_index(s, lane, q)
                k = i % p
                u = np.array([v[(lane, q, rr)] * np.exp(-2j * np.pi * rr * k * (g.m // (p * r)) / g.m) for rr in range(r)])
                y = np.fft.fft(u)
                for rr in range(r):
                    out[(lane, q, rr)] = y[rr]
        if s == g.ns - 1:
            res = np.zeros((g.lines, g.m), complex)
            for lane in range(g.team):
                for q in range(g.q[s]):
                    l, i, ok = g.lane_index(s, lane, q)
                    if ok:
                        for rr in range(r):
                            res[l, i + rr * p] = out[(lane, q, rr)]
            return float(np.abs(res - np.fft.fft(x, axis=1)).max())
        buf = {}
        block = p * r
        for lane in range(g.team):
            for q in range(g.q[s]):
                l, i, ok = g.lane_index(s, lane, q)
                if ok:
                    for rr in range(r):
                        a = l * pitch + phys((i // p) * block + (i % p) + rr * p, block, pads[s])
                        assert a not in buf and a < g.lines * pitch
                        buf[a] = out[(lane, q, rr)]
        v = {}
        r2, t2 = g.r[s + 1], g.t[s + 1]
        for lane in range(g.team):
            for q in range(g.q[s + 1]):
                l, i, ok = g.lane_index(s + 1, lane, q)
                for rr in range(r2):
                    v[(lane, q, rr)] = buf[l * pitch + phys(i + rr * t2, block, pads[s])] if ok else 0.0
    raise AssertionError


def geometry_for(m, kind):
    """-> (lines one batch owns, lanes of the team that works on a batch).
    rows: L * M <= 2048 on one wave (<= 32 complex values per lane; 3072 points: 48).  (A team of two waves for the
          3072-point rows -- 32 values per lane, no spills -- was 2x SLOWER: every team is a workgroup with its own copy of
          the 36 KB Bluestein tables in LDS, which leaves 2 teams = 4 waves per CU instead of 8 waves sharing one copy.)
    analysis columns: lengths >= 512 on a team of four waves with as many adjacent columns (<= 16: 128-byte row segments) as
          keep <= 36 values per lane (1080 rows: 8 columns, 34 per lane -- one wave owning 4 columns needed 72 per lane and
          the whole register file of a SIMD); shorter ones on one wave, <= 48 per lane.
    synthesis / plain columns: one wave, <= 48 values per lane (the synthesis workgroup is the four bands of the same
          columns and meets at its own barriers)."""
    if kind == "ROW":
        l = 1
        while l < 64 and 2 * l * m <= 2048:
            l *= 2
        return l, WAVE
    team = 256 if (kind == "COL" and m >= 512) else WAVE
    cap = 36 if team > WAVE else 48
    l = 1
    while l < 16 and 2 * l * m <= cap * team:
        l *= 2
    return l, team


def blu_capable(m):
    while m % 2 == 0:
        m //= 2
    return m in (1, 3)


ROW_LENGTHS = [3072, 2048, 1920, 1536, 1280, 1024, 960, 768, 640, 512, 480, 384, 320, 256, 240, 192, 160, 128, 120, 96, 80, 64, 60, 48,
               45, 40, 32, 30, 24, 20, 16, 15, 12]
COL_LENGTHS = [1536, 1080, 1024, 768, 720, 540, 512, 384, 360, 270, 256, 192, 180, 135, 128, 96, 90, 64, 48, 45, 32, 24, 16, 12]


def main():
    rng = np.random.default_rng(0)
    rows = []
    print("// GENERATED by tools/gen_wfft_configs.py -- do not edit; regenerate after changing the length lists there.")
    print("// Geometry of the wave-private FFT engine (vfi_wfft.h): X(M, L, TEAM, PITCH, PAD_0, PAD_1, PAD_2, radices...) for the row passes")
    print("// (ROW), the analysis column pass (COL) and the synthesis column pass (SYN)")
    print("//   M = engine length, L = lines (rows / adjacent columns) of one batch, TEAM = lanes that work on a batch (64 = one wave),")
    print("//   PITCH = dwords between the lines of the")
    print("//   exchange buffer, PAD_s = pad dwords per block of P_(s+1) positions in the exchange after stage s.")
    print("#pragma once")
    for cols, lengths, name in ((False, ROW_LENGTHS, "ROW"), (True, COL_LENGTHS, "COL"), (True, COL_LENGTHS, "SYN")):
        entries = []
        for m in lengths:
            lines, team = geometry_for(m, name)
            seq = choose_radices(m, lines, cols, blu_capable(m), team)
            g = Geometry(m, lines, cols, seq, team)
            pads, pitch, cyc = choose_layout(g)
            if name == "SYN" and lines * pitch < g.e * WAVE:      # (the synthesis columns also use the buffer as E x 64 scratch words)
                pitch += ((g.e * WAVE - lines * pitch + lines - 1) // lines + 31) // 32 * 32
            err = simulate(g, pads, pitch, rng)
            assert err < 1e-9, (m, seq, err)
            ideal = sum(q * r for q, r in zip(g.q, g.r)) - g.q[0] * g.r[0] if g.ns > 1 else 0
            rows.append((name, m, lines, seq, pads, pitch, g.e, cyc, err))
            pads4 = (list(pads) + [0, 0, 0])[:3]
            rad4 = (list(seq) + [1, 1, 1, 1])[:4]
            entries.append(f"    X({m}, {lines}, {team}, {pitch}, {pads4[0]}, {pads4[1]}, {pads4[2]}, {rad4[0]}, {rad4[1]}, {rad4[2]}, {rad4[3]})"
                           f"   /* E = {g.e}, stages {g.ns}, simulated LDS cycles per exchange set {cyc} */")
        print(f"#define VFI_WFFT_{name}_CONFIGS(X) \\")
        print(" \\\n".join(entries))
    for r in rows:
        print("//", r, file=sys.stderr)


if __name__ == "__main__":
    main()
