#!/usr/bin/env python3
"""Generator of csrc/vfi_conv_winograd4m_body.h: the hand-scheduled chunk body of the F(4x4,3x3) Winograd kernel with
M = 32 output channels per wave (one wave per SIMD, accumulators in the AGPR file).

One *chunk* = 4 input channels of one 16 x 64 output tile x 32 output channels.  Wave w owns tile row w (sixteen 4x4
tiles); lane (n16 = lane & 15, k4 = lane >> 4) holds the raw 6x6 patch of tile n16 for input channel k4, transforms it
(V = B^T d B, 144 operations) and feeds V's 36 entries to 72 v_mfma_f32_16x16x4_f32 (36 positions x two 16-channel
halves), i.e. 2 transform operations per MFMA instead of the 4 of the M = 16 kernel (vfi_conv_winograd4.hip).

The body of chunk c is ONE asm statement that
  * issues the 72 MFMAs of chunk c (operands: V rows from registers, weights U from registers),
  * in their shadow: reads chunk c+1's patch and weights from LDS, runs chunk c+1's column pass (B^T d), the row
    passes of chunk c (row i+1 during row i's MFMAs) and chunk c+1's row 0, and requests chunk c+4 from memory
    (LDS-DMA, buffer_load ... lds) into the ring slot chunk c vacated.
Every vector register the body touches is a FIXED physical register; the C++ side binds its variables to exactly these
registers with "{vN}" / "{a[N:M]}" constraints, so the compiler knows what is live where (see the operand macro emitted
at the end) and the text below can name registers literally.

Two bodies are emitted (P = 0 / 1): the patch sets swap roles every chunk.  tools/sim_wino4m.py interprets the emitted
text on the CPU (numpy over the 64 lanes) against a direct Winograd evaluation; tests/test_conv_host.py runs it.
"""
import os
import sys

ROWP, PLANE_S, IN_FLOATS, W_FLOATS, BN, NGRP = 68, 1280, 5120, 4608, 32, 9
BUF_BYTES = (IN_FLOATS + W_FLOATS) * 4
NBUF = 4

# ---- register map (VGPR numbers) -------------------------------------------------------------------------------------
TMP = [58, 60, 62, 64]          # column-pass temporaries: four even-aligned register PAIRS (v58..v65)
RPK = TMP[:3]                   # packed row pass: shares the pairs (passes run as whole blocks, never interleaved)
RTMP = [64, 65]                 # (unpacked variant: row-pass temporaries)
PADDR, WADDR = 66, 67           # LDS byte addresses of this chunk's reads (patch / weights)
TSET = [68, 104]                # patch sets A, B: row r at base + 6 r (6 registers)
UBASE = 140                     # u[h][g] (4 positions of group g, 16-channel half h) at UBASE + 4 (9 h + g)
VBUF = [212, 218]               # V rows: buffer 0 / 1 (6 registers each)
ACCV = 224                      # accumulators 64..71 (VGPRs); 0..63 are a[0:255]
FIRST_FREE = 58                 # the compiler keeps to v0 .. v57 inside the chunk loop
# ---- pinned SGPRs: the request cursor lives in the body (its per-chunk arithmetic runs in the MFMAs' shadow) ----------
S_K5, S_K41 = 86, 88                            # s[86:87] = (5, 5), s[88:89] = (4, 1): written at the head of every statement
S_RIN, S_RW, S_DMA, S_RD = 76, 80, 84, 85      # s[76:79] input descriptor, s[80:83] weight descriptor, LDS slot of the requests / reads


def treg(s, r, j):
    return TSET[s] + 6 * r + j


def ureg(h, p):
    return UBASE + 4 * (9 * h + p // 4) + p % 4


def acc_name(h, p):
    k = 36 * h + p
    if k < 64:
        return "a[%d:%d]" % (4 * k, 4 * k + 3)
    return "v[%d:%d]" % (ACCV + 4 * (k - 64), ACCV + 4 * (k - 64) + 3)


# Elimination builds (development aid, timing only -- results are garbage): W4M_ELIM=dma,barrier,vmcnt,mfma,valu,lds drops
# that class of instruction from the bodies.
ELIM = set(filter(None, os.environ.get("W4M_ELIM", "").split(",")))


class Emitter:
    """Instruction list with the LDS-read bookkeeping: a load's destination registers are pending until a
    `s_waitcnt lgkmcnt(k)` that retires it; k = number of LDS operations issued after it (in-order return)."""

    def __init__(self, pending_in=()):
        self.lines = []
        self.pending = [set(regs) for regs in pending_in]      # oldest first, one entry per outstanding LDS read
        self.lds_ops = 0

    def raw(self, text):
        op = text.split()[0]
        if ("dma" in ELIM and (op.startswith("buffer_load") or "m0" in text)) or ("barrier" in ELIM and op == "s_barrier") or \
           ("vmcnt" in ELIM and "vmcnt" in text) or ("mfma" in ELIM and op.startswith("v_mfma")) or \
           ("valu" in ELIM and op in ("v_fma_f32", "v_add_f32", "v_sub_f32", "v_pk_fma_f32", "v_pk_add_f32")) or ("lds" in ELIM and op.startswith("ds_read")):
            return
        self.lines.append(text)

    def need(self, regs):
        regs = set(regs)
        last = -1
        for i, p in enumerate(self.pending):
            if p & regs:
                last = i
        if last >= 0:
            k = len(self.pending) - 1 - last
            self.raw("s_waitcnt lgkmcnt(%d)" % min(k, 15))
            self.pending = self.pending[last + 1:] if k <= 15 else self.pending[len(self.pending) - 15:]

    def valu(self, text, reads, writes):
        self.need(list(reads) + list(writes))
        self.raw(text)

    def ds_read(self, width, dst, addr, offset):
        n = width // 32
        regs = list(range(dst, dst + n))
        self.need(regs)
        self.raw("ds_read_b%d v[%d:%d], v%d offset:%d" % (width, dst, dst + n - 1, addr, offset))
        self.pending.append(set(regs))

    def mfma(self, h, p, vb, first=False):
        a, b = ureg(h, p), vb
        self.need([a, b])
        acc = acc_name(h, p)
        self.raw("v_mfma_f32_16x16x4_f32 %s, v%d, v%d, %s" % (acc, a, b, "0" if first else acc))


def transform_ops(x, out, tmp, five):
    """B^T x for six samples (12 operations, the order of vfi_conv_winograd4.hip: input_transform6) -- x: input
    registers, out: output registers (may alias x only where IN_PLACE order allows, see column_pass), tmp: temporaries."""
    raise NotImplementedError


def column_pass_ops(s, j, five):
    """In place on column j of patch set s: 12 operations, 4 temporaries, no moves (the order matters: an output
    overwrites an input only after that input's last use)."""
    x = [treg(s, r, j) for r in range(6)]
    A, B, C, D = 60, 61, 62, 63
    ops = [
        ("v_fma_f32 v%d, -%s, v%d, v%d" % (C, five, x[2], x[4]), [x[2], x[4]], [C]),
        ("v_fma_f32 v%d, 4.0, v%d, v%d" % (x[0], x[0], C), [x[0], C], [x[0]]),                     # t0
        ("v_fma_f32 v%d, -4.0, v%d, v%d" % (A, x[2], x[4]), [x[2], x[4]], [A]),                    # p
        ("v_fma_f32 v%d, -4.0, v%d, v%d" % (B, x[1], x[3]), [x[1], x[3]], [B]),                    # q
        ("v_sub_f32 v%d, v%d, v%d" % (C, x[4], x[2]), [x[4], x[2]], [C]),                          # r
        ("v_sub_f32 v%d, v%d, v%d" % (D, x[3], x[1]), [x[3], x[1]], [D]),                          # s
        ("v_fma_f32 v%d, -%s, v%d, v%d" % (x[5], five, x[3], x[5]), [x[3], x[5]], [x[5]]),
        ("v_fma_f32 v%d, 4.0, v%d, v%d" % (x[5], x[1], x[5]), [x[1], x[5]], [x[5]]),               # t5
        ("v_add_f32 v%d, v%d, v%d" % (x[1], A, B), [A, B], [x[1]]),                                # t1
        ("v_sub_f32 v%d, v%d, v%d" % (x[2], A, B), [A, B], [x[2]]),                                # t2
        ("v_fma_f32 v%d, 2.0, v%d, v%d" % (x[3], D, C), [D, C], [x[3]]),                           # t3
        ("v_fma_f32 v%d, -2.0, v%d, v%d" % (x[4], D, C), [D, C], [x[4]]),                          # t4
    ]
    return ops


def row_pass_ops(s, i, vb, five):
    """Row i of patch set s (after its column pass) -> V row in buffer vb: 12 operations, 2 temporaries."""
    x = [treg(s, i, j) for j in range(6)]
    v = [vb + j for j in range(6)]
    E, F = RTMP
    ops = [
        ("v_fma_f32 v%d, -%s, v%d, v%d" % (v[0], five, x[2], x[4]), [x[2], x[4]], [v[0]]),
        ("v_fma_f32 v%d, 4.0, v%d, v%d" % (v[0], x[0], v[0]), [x[0], v[0]], [v[0]]),               # v0
        ("v_fma_f32 v%d, -4.0, v%d, v%d" % (E, x[2], x[4]), [x[2], x[4]], [E]),                    # p
        ("v_fma_f32 v%d, -4.0, v%d, v%d" % (v[2], x[1], x[3]), [x[1], x[3]], [v[2]]),              # q
        ("v_add_f32 v%d, v%d, v%d" % (v[1], E, v[2]), [E, v[2]], [v[1]]),                          # v1 = p + q
        ("v_sub_f32 v%d, v%d, v%d" % (v[2], E, v[2]), [E, v[2]], [v[2]]),                          # v2 = p - q
        ("v_sub_f32 v%d, v%d, v%d" % (F, x[4], x[2]), [x[4], x[2]], [F]),                          # r
        ("v_sub_f32 v%d, v%d, v%d" % (v[4], x[3], x[1]), [x[3], x[1]], [v[4]]),                    # s
        ("v_fma_f32 v%d, 2.0, v%d, v%d" % (v[3], v[4], F), [v[4], F], [v[3]]),                     # v3
        ("v_fma_f32 v%d, -2.0, v%d, v%d" % (v[4], v[4], F), [v[4], F], [v[4]]),                    # v4
        ("v_fma_f32 v%d, -%s, v%d, v%d" % (v[5], five, x[3], x[5]), [x[3], x[5]], [v[5]]),
        ("v_fma_f32 v%d, 4.0, v%d, v%d" % (v[5], x[1], v[5]), [x[1], v[5]], [v[5]]),               # v5
    ]
    return ops


# Packed forms (v_pk_fma_f32 / v_pk_add_f32: two fp32 lanes per instruction).  An inline constant (4.0, 2.0) sits in the
# low half of a packed source only, so op_sel_hi = 0 for that source makes both lanes read it; 5 and the pair (4, 1) are
# SGPR pairs.  On gfx950 an fp32 MFMA and the vector ALU
# do not overlap -- measured: a body with only its 72 MFMAs takes 2392 cycles, with the 144 scalar transform operations
# spread between them 3471, grouped 3056 -- so every transform INSTRUCTION is paid in full and halving their number is the
# lever.  Same fused operations, same operands as the scalar forms: bit-identical results.
PK = os.environ.get("W4M_PK", "1") != "0"
VMAP = [0, 2, 3, 4, 5, 1]        # packed row pass: V row register of column j (pairs (v0,v5), (v1,v2), (v3,v4))


def vreg(buf, j):
    return buf + (VMAP[j] if PK else j)


def pair(r):
    return "v[%d:%d]" % (r, r + 1)


def column_pass_pk(s, j):
    """Columns j, j+1 (j even) of patch set s, in place: the 12 operations of column_pass_ops on register pairs."""
    x = [treg(s, r, j) for r in range(6)]
    A, B, C, D = TMP
    P = pair
    two = lambda r: [r, r + 1]
    ops = [
        ("v_pk_fma_f32 %s, s[86:87], %s, %s neg_lo:[1,0,0] neg_hi:[1,0,0]" % (P(C), P(x[2]), P(x[4])), two(x[2]) + two(x[4]), two(C)),
        ("v_pk_fma_f32 %s, 4.0, %s, %s op_sel_hi:[0,1,1]" % (P(x[0]), P(x[0]), P(C)), two(x[0]) + two(C), two(x[0])),
        ("v_pk_fma_f32 %s, 4.0, %s, %s op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" % (P(A), P(x[2]), P(x[4])), two(x[2]) + two(x[4]), two(A)),
        ("v_pk_fma_f32 %s, 4.0, %s, %s op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" % (P(B), P(x[1]), P(x[3])), two(x[1]) + two(x[3]), two(B)),
        ("v_pk_add_f32 %s, %s, %s neg_lo:[0,1] neg_hi:[0,1]" % (P(C), P(x[4]), P(x[2])), two(x[4]) + two(x[2]), two(C)),
        ("v_pk_add_f32 %s, %s, %s neg_lo:[0,1] neg_hi:[0,1]" % (P(D), P(x[3]), P(x[1])), two(x[3]) + two(x[1]), two(D)),
        ("v_pk_fma_f32 %s, s[86:87], %s, %s neg_lo:[1,0,0] neg_hi:[1,0,0]" % (P(x[5]), P(x[3]), P(x[5])), two(x[3]) + two(x[5]), two(x[5])),
        ("v_pk_fma_f32 %s, 4.0, %s, %s op_sel_hi:[0,1,1]" % (P(x[5]), P(x[1]), P(x[5])), two(x[1]) + two(x[5]), two(x[5])),
        ("v_pk_add_f32 %s, %s, %s" % (P(x[1]), P(A), P(B)), two(A) + two(B), two(x[1])),
        ("v_pk_add_f32 %s, %s, %s neg_lo:[0,1] neg_hi:[0,1]" % (P(x[2]), P(A), P(B)), two(A) + two(B), two(x[2])),
        ("v_pk_fma_f32 %s, 2.0, %s, %s op_sel_hi:[0,1,1]" % (P(x[3]), P(D), P(C)), two(D) + two(C), two(x[3])),
        ("v_pk_fma_f32 %s, 2.0, %s, %s op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" % (P(x[4]), P(D), P(C)), two(D) + two(C), two(x[4])),
    ]
    return ops


def row_pass_pk(s, i, vb):
    """Row i of patch set s -> V row in buffer vb as the pairs (v0,v5), (v1,v2), (v3,v4): 6 packed operations."""
    x01, x23, x45 = treg(s, i, 0), treg(s, i, 2), treg(s, i, 4)
    I, PR, QS = RPK
    O0, O1, O2 = vb, vb + 2, vb + 4
    P = pair
    two = lambda r: [r, r + 1]
    ops = [
        ("v_pk_fma_f32 %s, s[86:87], %s, %s neg_lo:[1,0,0] neg_hi:[1,0,0]" % (P(I), P(x23), P(x45)), two(x23) + two(x45), two(I)),          # (-5 x2 + x4, -5 x3 + x5)
        ("v_pk_fma_f32 %s, 4.0, %s, %s op_sel_hi:[0,1,1]" % (P(O0), P(x01), P(I)), two(x01) + two(I), two(O0)),                                           # (v0, v5)
        ("v_pk_fma_f32 %s, s[88:89], %s, %s op_sel:[0,0,0] op_sel_hi:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]" % (P(PR), P(x23), P(x45)),
         two(x23) + two(x45), two(PR)),                                                                                                    # (p, r) = (x4 - 4 x2, x4 - x2)
        ("v_pk_fma_f32 %s, s[88:89], %s, %s op_sel:[0,1,1] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" % (P(QS), P(x01), P(x23)),
         two(x01) + two(x23), two(QS)),                                                                                                    # (q, s) = (x3 - 4 x1, x3 - x1)
        ("v_pk_add_f32 %s, %s, %s op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" % (P(O1), P(PR), P(QS)), two(PR) + two(QS), two(O1)),         # (v1, v2) = (p + q, p - q)
        ("v_pk_fma_f32 %s, 2.0, %s, %s op_sel:[0,1,1] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" % (P(O2), P(QS), P(PR)), two(QS) + two(PR), two(O2)),   # (v3, v4) = (2 s + r, -2 s + r)
    ]
    return ops


def column_pass_all(s, five):
    if PK:
        return [op for j in (0, 2, 4) for op in column_pass_pk(s, j)]
    return [op for j in range(6) for op in column_pass_ops(s, j, five)]


def row_pass_any(s, i, vb, five):
    return row_pass_pk(s, i, vb) if PK else row_pass_ops(s, i, vb, five)


def border_fixup(em, s):
    """Tiles on the left / right image border fetch their rows in 16-byte pieces like every other tile, so the halo column
    outside the image holds the neighbouring row's data (or 0 where the piece fell outside the buffer): the lanes whose
    patch contains such columns replace them here, in registers, before the column pass -- by 0 (zero padding: every
    column outside the image) or by the mirrored column (reflect padding: column -1 <- 1, column W <- W - 2; columns
    beyond W only feed outputs that are never stored).  %[fixmask]: bit j = column j of this lane's patch is replaced.
    Skipped (one scalar branch) for tiles that need nothing."""
    regs = [treg(s, r, j) for r in range(6) for j in range(6)]
    em.need(regs)                                              # (the wait must not sit inside the skipped block)
    A = 60
    em.raw("s_bitcmp1_b32 %[s_rflags], 1")
    em.raw("s_cbranch_scc0 5f")
    # One piece of a chunk cannot be fetched where it belongs: columns -1..2 of image row 0 of the chunk's FIRST channel in
    # a tile at the left border start 4 bytes before the buffer, and a 16-byte request whose first dword lies before the
    # base returns zeros for all four (measured, tools/probes/buffer_oob.hip).  The kernel fetches columns 0..3 there
    # instead; the one lane and patch row that read them (%[fixmask] bit 8 + row) move them up by one column.
    for i in range(6):
        em.raw("v_and_b32 v%d, %d, %%[fixmask]" % (A, 256 << i))
        em.raw("v_cmp_ne_u32 vcc, 0, v%d" % A)
        for j in (3, 2, 1):
            em.raw("v_cndmask_b32 v%d, v%d, v%d, vcc" % (treg(s, i, j), treg(s, i, j), treg(s, i, j - 1)))
    em.raw("s_bitcmp1_b32 %[s_rflags], 2")
    em.raw("s_cbranch_scc1 4f")
    for j in range(6):                                         # zero padding
        em.raw("v_and_b32 v%d, %d, %%[fixmask]" % (A, 1 << j))
        em.raw("v_cmp_ne_u32 vcc, 0, v%d" % A)
        for r in range(6):
            em.raw("v_cndmask_b32 v%d, v%d, 0, vcc" % (treg(s, r, j), treg(s, r, j)))
    em.raw("s_branch 5f")
    em.raw("4:")
    for j in (5, 4, 3, 2, 0):                                  # reflect padding
        em.raw("v_and_b32 v%d, %d, %%[fixmask]" % (A, 1 << j))
        em.raw("v_cmp_ne_u32 vcc, 0, v%d" % A)
        for r in range(6):
            em.raw("v_cndmask_b32 v%d, v%d, v%d, vcc" % (treg(s, r, j), treg(s, r, j), treg(s, r, 2 if j == 0 else j - 2)))
    em.raw("5:")


def patch_reads(em, s):
    for r in range(6):
        em.ds_read(128, treg(s, r, 0), PADDR, r * ROWP * 4)
        em.ds_read(64, treg(s, r, 4), PADDR, r * ROWP * 4 + 16)


def weight_read(em, h, g):
    em.ds_read(128, UBASE + 4 * (9 * h + g), WADDR, (g * BN + h * 16) * 16)


def const_setup(em):
    """5.0 and the pair (4, 1) of the packed transforms: two scalar moves each at the head of the statement instead of four
    scalar registers that live across the whole loop (the kernel is short of them: every one that spills costs a
    v_readlane per chunk)."""
    if PK:
        em.raw("s_mov_b32 s%d, 0x40a00000" % S_K5)
        em.raw("s_mov_b32 s%d, 0x40a00000" % (S_K5 + 1))
        em.raw("s_mov_b32 s%d, 0x40800000" % S_K41)
        em.raw("s_mov_b32 s%d, 0x3f800000" % (S_K41 + 1))


def addr_setup(em):
    em.raw("v_add_u32 v%d, s%d, %%[pa0]" % (PADDR, S_RD))
    em.raw("v_add_u32 v%d, s%d, %%[wa0]" % (WADDR, S_RD))


def read_cursor_step(em):
    """The ring slot the NEXT statement reads from."""
    em.raw("s_add_u32 s%d, s%d, %d" % (S_RD, S_RD, BUF_BYTES))
    em.raw("s_cmp_ge_u32 s%d, %d" % (S_RD, NBUF * BUF_BYTES))
    em.raw("s_cbranch_scc0 7f")
    em.raw("s_mov_b32 s%d, 0" % S_RD)
    em.raw("7:")


def request_cursor_step(em):
    """Past the chunk just requested: descriptor bases move on by one chunk, the remaining input bytes shrink (saturating:
    the channel tail of the last chunk and everything past the item read as 0), next ring slot.  At an item boundary the
    kernel overwrites all of it after the statement."""
    em.raw("s_add_u32 s%d, s%d, %%[s_inc]" % (S_RIN, S_RIN))
    em.raw("s_addc_u32 s%d, s%d, 0" % (S_RIN + 1, S_RIN + 1))
    em.raw("s_sub_u32 s%d, s%d, %%[s_inc]" % (S_RIN + 2, S_RIN + 2))
    em.raw("s_cselect_b32 s%d, 0, s%d" % (S_RIN + 2, S_RIN + 2))
    em.raw("s_add_u32 s%d, s%d, %%[s_winc]" % (S_RW, S_RW))
    em.raw("s_addc_u32 s%d, s%d, 0" % (S_RW + 1, S_RW + 1))
    em.raw("s_add_u32 s%d, s%d, %d" % (S_DMA, S_DMA, BUF_BYTES))
    em.raw("s_cmp_ge_u32 s%d, %%[s_dma_end]" % S_DMA)
    em.raw("s_cbranch_scc0 6f")
    em.raw("s_sub_u32 s%d, s%d, %d" % (S_DMA, S_DMA, NBUF * BUF_BYTES))
    em.raw("6:")


def dma_piece(em, kind, t):
    """One 1 KiB LDS-DMA request of chunk c+4: input piece t (0..4) or weight piece t (0..4) of this wave."""
    if kind == "in":
        em.raw("s_add_i32 m0, s%d, %d" % (S_DMA, 4096 * t))
        em.raw("s_nop 0")
        em.raw("buffer_load_dwordx4 %%[voff%d], s[%d:%d], 0 offen lds" % (t, S_RIN, S_RIN + 3))
    else:
        if t == 4:
            em.raw("s_cmp_lt_u32 %[s_wave], 2")               # weight pieces 16, 17 exist for waves 0, 1 only
            em.raw("s_cbranch_scc0 1f")
        em.raw("s_add_i32 m0, s%d, %d" % (S_DMA, IN_FLOATS * 4 + 4096 * t))
        em.raw("s_nop 0")
        em.raw("buffer_load_dwordx4 %%[wvoff], s[%d:%d], %s offen lds" % (S_RW, S_RW + 3, "%[s_w0]" if t == 0 else "%[s_wt]"))
        if t == 4:
            em.raw("1:")
        else:                                                  # the next piece's offset, long before its request reads it
            em.raw("s_add_u32 %%[s_wt], %s, %%[s_wstep]" % ("%[s_w0]" if t == 0 else "%[s_wt]"))


def entry_wait(em, chunks_in_flight):
    """Own requests of the chunk about to be read have landed once only those of the `chunks_in_flight` younger chunks
    are outstanding (waves 0, 1 issue 10 per chunk, waves 2, 3 nine); then the workgroup barrier publishes everyone's."""
    em.raw("s_cmp_lt_u32 %[s_wave], 2")
    em.raw("s_cbranch_scc1 2f")
    em.raw("s_waitcnt vmcnt(%d)" % (9 * chunks_in_flight))
    em.raw("s_branch 3f")
    em.raw("2:")
    em.raw("s_waitcnt vmcnt(%d)" % (10 * chunks_in_flight))
    em.raw("3:")
    em.raw("s_waitcnt lgkmcnt(0)")                             # (this wave's reads of the slot the DMA is about to reuse)
    em.pending = []
    em.raw("s_barrier")


def gen_prime():
    """Before the first chunk: chunk 0 has landed -> patch set 0 and the weights into registers, column pass, row 0."""
    em = Emitter()
    five = "%[s_five]"
    const_setup(em)
    addr_setup(em)
    entry_wait(em, 3)
    patch_reads(em, 0)
    for h in range(2):
        for g in range(NGRP):
            weight_read(em, h, g)
    border_fixup(em, 0)
    for text, rd, wr in column_pass_all(0, five):
        em.valu(text, rd, wr)
    for text, rd, wr in row_pass_any(0, 0, VBUF[0], five):
        em.valu(text, rd, wr)
    em.raw("s_waitcnt lgkmcnt(0)")
    read_cursor_step(em)
    return em.lines, []


# gap (MFMA index) behind which the next chunk's first column-pass block sits; the row pass of row 2 sits behind MFMA 18: with
# 18 the two form ONE block of 18 operations (every switch between the matrix pipe and the vector ALU costs ~15 cycles)
COLGAP = int(os.environ.get("W4M_COLGAP", "18"))


def mfma_order():
    """(row i, column j, half h) in issue order: 12 per row of positions."""
    return [(i, j, h) for i in range(6) for j in range(6) for h in range(2)]


def gen_body(P, pending_in, first=False):
    """first: the item's first chunk -- the MFMAs start from 0 instead of the accumulators (nothing has to clear them)."""
    cur, nxt = P, 1 - P
    five = "%[s_five]"
    em = Emitter(pending_in)
    order = mfma_order()
    # last use (gap index) of every weight group / half
    last_use = {}
    for m, (i, j, h) in enumerate(order):
        last_use[(h, (6 * i + j) // 4)] = m
    # ---- work lists, each entry (earliest gap, callable) ----
    sched = {g: [] for g in range(72)}

    cluster = int(os.environ.get("W4M_CLUSTER", "6"))      # transform operations go in blocks behind every cluster-th MFMA (see PK above)

    def at(gap, fn, cost, valu=False):
        if valu and cluster > 1:
            gap = min(71, gap // cluster * cluster + cluster - 1)
        sched[gap].append((fn, cost))

    # patch reads of chunk c+1: gaps 1..6, two per gap (behind the barrier that follows MFMA 0)
    k = 0
    for r in range(6):
        at(1 + k // 2, (lambda r=r: em.ds_read(128, treg(nxt, r, 0), PADDR, r * ROWP * 4)), 1)
        k += 1
        at(1 + k // 2, (lambda r=r: em.ds_read(64, treg(nxt, r, 4), PADDR, r * ROWP * 4 + 16)), 1)
        k += 1
    # weight reloads: one per gap from the gap after the last use (never beside the patch reads' gaps 1..6)
    busy = {g: 0 for g in range(80)}
    for g in range(1, 7):
        busy[g] = 2
    wl = sorted(last_use.items(), key=lambda kv: kv[1])
    tail = []
    for (h, g), m in wl:
        gap = m
        while gap < 72 and busy[gap] >= 1:
            gap += 1
        if gap >= 72:
            tail.append((h, g))
        else:
            busy[gap] += 1
            at(gap, (lambda h=h, g=g: weight_read(em, h, g)), 1)
    # row passes of chunk c: row i+1 during the first six gaps of row i
    valu = lambda text, rd, wr: (lambda: em.valu(text, rd, wr))
    if PK:
        # The transform operations go in a few blocks (an fp32 MFMA and a VALU operation never overlap; every switch between
        # the two costs).  Inside a block two INDEPENDENT passes alternate instruction by instruction -- a row pass of this
        # chunk and a piece of the next chunk's column pass -- so that a packed operation's result is not needed by the very
        # next instruction.  W4M_MIX=0: passes as separate blocks (A/B).
        mix = False         # (measured: interleaving a row pass with a column pass inside a block gains 1 % and costs six registers)
        rows = [row_pass_pk(cur, i + 1, VBUF[(i + 1) % 2]) for i in range(5)]
        cols = [op for jj in range(3) for op in column_pass_pk(nxt, 2 * jj)]
        at(13, (lambda: border_fixup(em, nxt)), 2)
        for text, rd, wr in rows[0]:
            at(6, valu(text, rd, wr), 1)
        if mix:
            for b in range(4):                                  # blocks behind MFMA 18, 30, 42, 54: row b+2 with 9 column operations
                r, c = list(rows[b + 1]), cols[9 * b:9 * b + 9]
                while r or c:
                    if c:
                        at(18 + 12 * b, valu(*c.pop(0)), 1)
                    if r:
                        at(18 + 12 * b, valu(*r.pop(0)), 1)
                    if c and len(c) > len(r):
                        at(18 + 12 * b, valu(*c.pop(0)), 1)
        else:
            for i in range(1, 5):
                for text, rd, wr in rows[i]:
                    at(12 * i + 6, valu(text, rd, wr), 1)
            for jj in range(3):
                for text, rd, wr in cols[12 * jj:12 * jj + 12]:
                    at(COLGAP + 12 * jj, valu(text, rd, wr), 1)
        for text, rd, wr in row_pass_pk(nxt, 0, VBUF[0]):
            at(64, valu(text, rd, wr), 1)
    else:
        for i in range(5):
            for n, (text, rd, wr) in enumerate(row_pass_ops(cur, i + 1, VBUF[(i + 1) % 2], five)):
                at(12 * i + n // 2, valu(text, rd, wr), 1, True)
        at(13, (lambda: border_fixup(em, nxt)), 2)
        n = 0
        for j in range(6):
            for text, rd, wr in column_pass_ops(nxt, j, five):
                at(14 + n // 2, valu(text, rd, wr), 1, True)
                n += 1
        for n, (text, rd, wr) in enumerate(row_pass_ops(nxt, 0, VBUF[0], five)):
            at(60 + n // 2, valu(text, rd, wr), 1, True)
    # DMA requests of chunk c+4
    dma_gaps = [8, 13, 20, 25, 32, 37, 44, 49, 56, 66]
    kinds = [("in", 0), ("w", 0), ("in", 1), ("w", 1), ("in", 2), ("w", 2), ("in", 3), ("w", 3), ("in", 4), ("w", 4)]
    for gap, (kind, t) in zip(dma_gaps, kinds):
        at(gap, (lambda kind=kind, t=t: dma_piece(em, kind, t)), 6)
    at(68, (lambda: request_cursor_step(em)), 2)          # (after the last request that uses the descriptors / the slot)
    at(30, (lambda: read_cursor_step(em)), 1)             # (after the statement's last use of the read slot: its address setup)

    # ---- emit ----
    const_setup(em)
    addr_setup(em)
    for m, (i, j, h) in enumerate(order):
        em.mfma(h, 6 * i + j, vreg(VBUF[i % 2], j), first)
        if m == 0:
            entry_wait(em, 2)
        for fn, _ in sched[m]:
            fn()
    for h, g in tail:
        weight_read(em, h, g)
    return em.lines, [sorted(p) for p in em.pending]


def c_string(lines):
    return "\n".join('    "%s\\n\\t"' % ln for ln in lines)


def operands_macro(kind):
    """Operand list of one statement: ST = the state struct (see vfi_conv_winograd4m.hip).  kind: "prime", 0 / 1 (body of
    that parity) or "drain".  Only what a statement really reads is an input and only what the NEXT statement needs is
    live afterwards: the patch set a body consumes is input-only, the set it fills output-only, so that set's registers
    (and the second V row, the temporaries) are free for the compiler between the statements (item epilogue)."""
    outs, ins = [], []
    for k in range(72):
        h, p = divmod(k, 36)
        outs.append('"+{%s}"(ST.acc[%d])' % (acc_name(h, p), k))
    if kind in ("drain", "zero"):
        return outs, ins

    def tset(s, fmt_lo, fmt_hi, dst):
        for r in range(6):
            b = treg(s, r, 0)
            dst.append(fmt_lo % (b, b + 3, s, r))
            dst.append(fmt_hi % (b + 4, b + 5, s, r))

    if kind == "prime":
        tset(0, '"=&{v[%d:%d]}"(ST.t_lo[%d][%d])', '"=&{v[%d:%d]}"(ST.t_hi[%d][%d])', outs)
        uc = '"=&{v[%d:%d]}"(ST.u[%d])'
        vc = ['"=&{v[%d:%d]}"(ST.vb[%d])'] * 3
    elif kind == "loop":                                     # pairs of chunks: set 0 comes in and goes out, set 1 is scratch
        tset(0, '"+{v[%d:%d]}"(ST.t_lo[%d][%d])', '"+{v[%d:%d]}"(ST.t_hi[%d][%d])', outs)
        tset(1, '"=&{v[%d:%d]}"(ST.t_lo[%d][%d])', '"=&{v[%d:%d]}"(ST.t_hi[%d][%d])', outs)
        uc = '"+{v[%d:%d]}"(ST.u[%d])'
        vc = ['"+{v[%d:%d]}"(ST.vb[%d])', '"+{v[%d:%d]}"(ST.vb[%d])', '"=&{v[%d:%d]}"(ST.vb[%d])']
    else:
        tset(1 - kind, '"=&{v[%d:%d]}"(ST.t_lo[%d][%d])', '"=&{v[%d:%d]}"(ST.t_hi[%d][%d])', outs)
        tset(kind, '"{v[%d:%d]}"(ST.t_lo[%d][%d])', '"{v[%d:%d]}"(ST.t_hi[%d][%d])', ins)
        uc = '"+{v[%d:%d]}"(ST.u[%d])'
        # V row buffer 0 = v212..217 carries row 0 into the next statement; it spans vb[0] and half of vb[1]
        vc = ['"+{v[%d:%d]}"(ST.vb[%d])', '"+{v[%d:%d]}"(ST.vb[%d])', '"=&{v[%d:%d]}"(ST.vb[%d])']
    for k in range(18):
        outs.append(uc % (UBASE + 4 * k, UBASE + 4 * k + 3, k))
    for k in range(3):
        outs.append(vc[k] % (VBUF[0] + 4 * k, VBUF[0] + 4 * k + 3, k))
    for k, r in enumerate(list(range(FIRST_FREE, PADDR)) + [PADDR, WADDR]):
        outs.append('"=&{v%d}"(ST.tmp[%d])' % (r, k))
    outs.append('"+{s%d}"(ST.s_rd)' % S_RD)
    ins += ['[pa0] "v"(ST.pa0)', '[wa0] "v"(ST.wa0)', '[fixmask] "v"(ST.fixmask)', '[s_rflags] "s"(ST.s_rflags)',
            '[s_wave] "s"(ST.s_wave)'] + ([] if PK else ['[s_five] "s"(ST.s_five)'])
    if PK:
        outs += ['"=&{s[%d:%d]}"(ST.k5)' % (S_K5, S_K5 + 1), '"=&{s[%d:%d]}"(ST.k41)' % (S_K41, S_K41 + 1)]
    if kind == "loop":
        outs.append('[s_n] "+s"(ST.s_n)')
    if kind != "prime":
        outs += ['"+{s[%d:%d]}"(ST.rin)' % (S_RIN, S_RIN + 3), '"+{s[%d:%d]}"(ST.rw)' % (S_RW, S_RW + 3), '"+{s%d}"(ST.s_dma)' % S_DMA]
        ins += ['[wvoff] "v"(ST.wvoff)'] + ['[voff%d] "v"(ST.voff[%d])' % (t, t) for t in range(5)]
        outs.append('[s_wt] "=&s"(ST.s_wt)')
        ins += ([] if kind == "loop" else ['[s_first] "s"(ST.s_first)']) + ['[s_inc] "s"(ST.s_inc)', '[s_winc] "s"(ST.s_winc)', '[s_dma_end] "s"(ST.s_dma_end)',
                '[s_w0] "s"(ST.s_w[0])', '[s_wstep] "s"(ST.s_wstep)']
    return outs, ins


def main(out_path):
    prime, _ = gen_prime()
    # the reads left in flight at the end of a body are the reads pending at the start of the next one: iterate to the fixed point
    pend = {0: [], 1: []}
    for _ in range(3):
        b0, t0 = gen_body(0, pend[0])
        b1, t1 = gen_body(1, pend[1])
        pend = {0: t1, 1: t0}
    b0, t0 = gen_body(0, pend[0])
    b1, t1 = gen_body(1, pend[1])
    assert t1 == pend[0] and t0 == pend[1], "tail reads did not reach a fixed point"
    b0f, b1f = gen_body(0, pend[0], True)[0], gen_body(1, pend[1], True)[0]
    n_mfma = sum(1 for ln in b0 if ln.startswith("v_mfma"))
    n_valu = sum(1 for ln in b0 if ln.startswith(("v_fma", "v_add_f32", "v_sub", "v_pk_")))
    assert ELIM or (n_mfma == 72 and n_valu == (72 if PK else 144)), (n_mfma, n_valu)
    with open(out_path, "w") as f:
        f.write("// GENERATED by tools/gen_wino4m.py -- do not edit.  Chunk body of conv3x3_winograd4m_kernel (vfi_conv_winograd4m.hip).\n")
        f.write("// per body: %d MFMAs, %d transform operations, %d LDS reads, 10 LDS-DMA requests\n" %
                (n_mfma, n_valu, sum(1 for ln in b0 if ln.startswith("ds_read"))))
        f.write("#pragma once\n")
        f.write("#define W4M_FIRST_FIXED_VGPR %d\n" % FIRST_FREE)
        f.write("#define W4M_ASM_PRIME \\\n" + c_string(prime).replace("\n", " \\\n") + "\n")
        f.write("// before the epilogue reads the accumulators: the last MFMAs (8 passes) must have written them back\n")
        f.write('#define W4M_ASM_DRAIN "s_nop 7\\n\\ts_nop 7\\n\\t"\n')
        zero = ["v_accvgpr_write_b32 a%d, 0" % n for n in range(256)] + ["v_mov_b32 v%d, 0" % (ACCV + n) for n in range(32)]
        f.write("// after the epilogue has read them (its reads name the registers literally: the accumulators never become C++ values,\n"
                "// or the compiler would copy all 288 into VGPRs)\n")
        f.write("#define W4M_ACCV %d\n" % ACCV)
        f.write("#define W4M_ASM_ZERO \\\n" + c_string(zero).replace("\n", " \\\n") + "\n")
        # One statement per parity holds BOTH forms of the chunk -- accumulating, and (an item's first chunk, %[s_first] != 0)
        # starting from srcC = 0, so nothing ever has to clear the 288 accumulator registers -- behind one scalar branch: two
        # statements merged by an if / else in C++ would need phi nodes of the pinned scalar outputs, which the compiler
        # cannot place ("illegal VGPR to SGPR copy").
        both = lambda normal, first: ["s_cmp_lg_u32 %[s_first], 0", "s_cbranch_scc1 8f"] + normal + ["s_branch 9f", "8:"] + first + ["9:"]
        f.write("#define W4M_ASM_BODY0 \\\n" + c_string(both(b0, b0f)).replace("\n", " \\\n") + "\n")
        f.write("#define W4M_ASM_BODY1 \\\n" + c_string(both(b1, b1f)).replace("\n", " \\\n") + "\n")
        # %[s_n] PAIRS of chunks (parity 0 then 1) of one item with no item boundary on any of the three cursors: the whole
        # loop control is two scalar instructions, where the compiler's code between two statements costs ~300 idle cycles
        loop = ["Lw4m_loop_%=:"] + b0 + b1 + ["s_sub_u32 %[s_n], %[s_n], 1", "s_cmp_lg_u32 %[s_n], 0", "s_cbranch_scc1 Lw4m_loop_%="]
        f.write("#define W4M_ASM_LOOP \\\n" + c_string(loop).replace("\n", " \\\n") + "\n")
        for kind, tag in (("prime", "PRIME"), (0, "BODY0"), (1, "BODY1"), ("loop", "LOOP"), ("drain", "DRAIN"), ("zero", "ZERO")):
            outs, ins = operands_macro(kind)
            f.write("#define W4M_OPERANDS_%s(ST) \\\n    : " % tag + ", \\\n      ".join(outs) + " \\\n    : " + ", \\\n      ".join(ins) + "\n")
    return prime, b0, b1, b0f, b1f


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    default = os.path.join(here, "..", "fusion-method-for-video-frame-interpolation_amd", "csrc", "vfi_conv_winograd4m_body.h")
    main(sys.argv[1] if len(sys.argv) > 1 else default)
