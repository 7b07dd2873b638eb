"""Generator of the hand-scheduled K loop of k_gemm_f16_w4 (csrc/vit.hip): writes csrc/gemm_w4_loop.inc.

    python vilgod_amd/csrc/gen_gemm_w4.py            # rewrite the .inc
    python vilgod_amd/csrc/gen_gemm_w4.py --check    # exit 1 if the committed .inc differs from what this script emits

What the block is (round 6; the projection GEMMs of third_party/CLIP/clip/model.py:171-192).  ONE assembly block = the whole K loop of
one 256 x 256 output tile of a PERSISTENT workgroup, including the first DMA pieces of the workgroup's NEXT tile:
  * workgroup = 4 waves, one per SIMD; 256 x 256 x 64 macro tile; wave (wm, wn) owns 128 tokens x 128 features = 8 x 8 tiles of
    v_mfma_f32_16x16x32_f16, i.e. 256 accumulator registers, which live in a0..a255 for the whole tile (the compiler never sees them:
    the epilogue fetches them with v_accvgpr_read).  A operand = 16 token rows (X fragment mi), B operand = 16 weight rows (W fragment
    ni): lane l of accumulator tile (ni, mi) holds tokens mi 16 + 4 (l >> 4) + e and the feature whose weight row sits at LDS row
    ni 16 + (l & 15) of the wave's half.  WHICH feature that is the kernel chooses through the W pieces' source addresses (the LDS
    image of W is a row permutation of the tile): for fp16 outputs LDS row ni 16 + r holds feature 8 r + ni, so a lane's eight tiles
    ni = 0..7 are eight consecutive features = one 16-byte store, sixteen lanes = 256 contiguous bytes of an output row; for fp32
    outputs feature 64 (ni >> 2) + 4 r + (ni & 3).  The epilogue therefore needs neither LDS nor barriers.
  * LDS: a ring of FIVE 32 KB slots (all 160 KB), a slot = 256 rows x 128 B of one operand for one K-tile, 16-byte chunk c of row r at
    c ^ ((r >> 1) & 7).  Filled by LDS-DMA (`buffer_load_dwordx4 ... lds`, a piece = 8 rows x 128 B, the swizzle applied to the
    per-lane SOURCE address); a wave fills rows [64 w, 64 w + 64) of every slot: 8 + 8 pieces per K-tile.  At iteration i the ring
    registers a..e hold the slots of X(i), W(i), X(i+1), W(i+1), X(i+2).
  * ONE barrier per K-tile ("M").  Iteration i: the first k32 sub-step's MFMAs run on fragments read during iteration i - 1 while the
    second sub-step's fragments are read; `s_waitcnt vmcnt(8) lgkmcnt(0)` + s_barrier: every wave is done with the slots of X(i), W(i)
    and every wave's pieces of X(i+1), W(i+1) have landed.  Behind M: W(i+2) goes into X(i)'s slot, X(i+3) into W(i)'s, and the first
    sub-step's fragments of tile i + 1 are read.  So W has one iteration (>= 2 048 cycles of MFMAs) to land and X two.
  * the last three iterations have nothing of this tile left to fetch: their DMA slots carry the NEXT output tile's X'(0), W'(0),
    X'(1), W'(1), X'(2) (source bases nxlo/nxhi, nwlo/nwhi), so the block leaves the ring exactly as the first tile's prologue does
    and the next block starts its MFMAs as soon as X'(0), W'(0) have landed.  The pieces are issued BEFORE the epilogue's stores:
    vector memory operations retire in order, so the next block can wait for them with a count that leaves the stores in flight
    (`vmcnt(24 + S)`, `vmcnt(8 + S)` at the first M; S = the epilogue's stores, a property of the kernel instantiation: ST below).
    The first iteration's MFMAs take 0 as their C operand: no zeroing pass over the accumulators.
  * every LDS read, DMA piece, address update and wait sits at a fixed distance between the MFMAs (the table SCHED below): nothing is
    left to the compiler's scheduler.  A DMA piece costs the CU's address path 16 cycles (1 KB per wave, 64 B per cycle) and the four waves
    issue theirs together: pieces closer than every third gap stall the issuing waves (measured), and X's pieces, which have two
    iterations to land, are spread to every sixth (in_proj -3 %, c_proj -5 % against every third).

Registers: v128..v255 fragments (W half 0, X half 0, W half 1, X half 1: 32 each), v120..v123 read addresses, s60..s91 (piece offsets,
slot ring, buffer descriptors, loop counter; s92..s101 cycle stamps of the trace variant).  Everything else comes in through named
operands: per-lane DMA offsets dv0/dv1 (X, even / odd pieces) and dw0/dw1 (W), fragment read offsets xo0/xo1/wo0/wo1, the operand
bases of this wave for this tile (xlo/xhi, wlo/whi) and the next (nxlo/nxhi, nwlo/nwhi), rowb = bytes per operand row, wpo = byte
offset of W's odd pieces, lds0, wdst, np (>= 4), first (1: nothing has been prefetched, run the prologue), ring (in/out: byte offset of
slot a).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, 'gemm_w4_loop.inc')

FW = (128, 192)          # W fragments (B operand), k32 half 0 / 1: v[FW[h] + 4 ni .. + 3]
FX = (160, 224)          # X fragments (A operand)
VA_W = (120, 121)        # LDS read address of the W fragments, half 0 / 1
VA_X = (122, 123)
S_OFF = {'x': 60, 'w': 84}     # s60..s66 / s84..s90: source offsets of pieces p = 1..7 (X: p x 8 rows; W: (p & 1) wpo + (p >> 1) rows)
S_DST = 67               # slot + this wave's share: M0 base of the pieces being issued
S_SLOT = (68, 69, 70, 71, 72)   # a, b, c, d, e = slots of X(i), W(i), X(i+1), W(i+1), X(i+2)
S_TMP = 73
S_CNT = 74
S_WDST = 75              # this wave's byte offset inside a slot (64 rows x 128 B x wave)
SRD = {'x': 76, 'w': 80}       # s[76:79], s[80:83]
S_END = 91               # lds0 + 5 slots (ring wrap)
S_OFF0 = {'x': 58, 'w': 59}    # IOFF: source offset of piece 0 (3 072: the descriptor bases sit that much low)

# vmcnt allowances by output kind: how many of the epilogue's vector-memory operations (its stores) may still be in flight when the
# next block waits for its first pieces.  'h': fp16 outputs, exactly 32 16-byte stores per lane; 'f': fp32 outputs, >= 64 (the count is
# capped by the 6-bit counter).  tests/test_abi.py checks the store counts of every instantiation against these.
ST = {'h': 32, 'f': 64}


def mfma(k, c0=False):
    h, kk = divmod(k, 64)
    mi, ni = divmod(kk, 8)         # the A operand (token rows) stays for eight MFMAs
    a = 4 * (ni * 8 + mi)
    c = '0' if (c0 and h == 0) else 'a[%d:%d]' % (a, a + 3)
    return 'v_mfma_f32_16x16x32_f16 a[%d:%d], v[%d:%d], v[%d:%d], %s' % (
        a, a + 3, FX[h] + 4 * mi, FX[h] + 4 * mi + 3, FW[h] + 4 * ni, FW[h] + 4 * ni + 3, c)


def rd(op, h, i):
    base = (FW if op == 'w' else FX)[h] + 4 * i
    addr = (VA_W if op == 'w' else VA_X)[h]
    return 'ds_read_b128 v[%d:%d], v%d offset:%d' % (base, base + 3, addr, 2048 * i)


def addr(op, h, slot):
    return 'v_add_u32 v%d, s%d, %%[%so%d]' % ((VA_W if op == 'w' else VA_X)[h], slot, op, h)


def dst(slot):
    return 's_add_u32 s%d, s%d, s%d' % (S_DST, slot, S_WDST)


IOFF = True     # pieces p & 3 != 0 reach their LDS rows through the instruction's 12-bit offset (added to the LDS address AND to the source address:
                # the descriptors' bases sit 3 072 bytes low and every piece's source offset carries 3 072 - 1 024 (p & 3)): M0 is written twice per
                # eight pieces instead of eight times


def m0(p):
    if IOFF:
        return None if p & 3 else ('s_mov_b32 m0, s%d' % S_DST if p == 0 else 's_add_u32 m0, s%d, 4096' % S_DST)
    return 's_mov_b32 m0, s%d' % S_DST if p == 0 else 's_add_u32 m0, s%d, %d' % (S_DST, 1024 * p)


def piece(op, p):
    if IOFF:
        so = 's%d' % (S_OFF0[op] if p == 0 else S_OFF[op] + p - 1)
        io = ' offset:%d' % (1024 * (p & 3)) if p & 3 else ''
        return 'buffer_load_dwordx4 %%[d%s%d], s[%d:%d], %s offen%s lds' % ('v' if op == 'x' else 'w', p & 1, SRD[op], SRD[op] + 3, so, io)
    so = '0' if p == 0 else 's%d' % (S_OFF[op] + p - 1)
    return 'buffer_load_dwordx4 %%[d%s%d], s[%d:%d], %s offen lds' % ('v' if op == 'x' else 'w', p & 1, SRD[op], SRD[op] + 3, so)


def advance(op, n=1):
    return ['s_add_u32 s%d, s%d, %d' % (SRD[op], SRD[op], 128 * n), 's_addc_u32 s%d, s%d, 0' % (SRD[op] + 1, SRD[op] + 1)]


def srd_base(op, nxt=False):
    lo, hi = ('n' if nxt else '') + op + 'lo', ('n' if nxt else '') + op + 'hi'
    if IOFF:
        return ['s_sub_u32 s%d, %%[%s], 3072' % (SRD[op], lo), 's_subb_u32 s%d, %%[%s], 0' % (SRD[op] + 1, hi)]
    return ['s_mov_b32 s%d, %%[%s]' % (SRD[op], lo), 's_mov_b32 s%d, %%[%s]' % (SRD[op] + 1, hi)]


def rotate():
    a, b, c, d, e = S_SLOT
    t = S_TMP
    return ['s_mov_b32 s%d, s%d' % p for p in ((t, a), (a, c), (c, e), (e, b), (b, d), (d, t))]


# Cycle stamps (trace variants only): s_memtime pairs around the waits and the barrier, differences accumulated in SGPRs.
# s[92:93] stamp, s94 = sum of M-wait cycles, s95 = barrier, s96 = end-of-iteration wait, s97 previous stamp, s98 scratch,
# s99 = cost of one stamp pair with nothing between (calibration), s100 loop entry stamp, s101 block entry stamp.
def stamp_first():
    return ['s_memtime s[92:93]']


def stamp_take(acc):
    """behind a wait that also retired the pending s_memtime: s97 = that stamp; new stamp; acc += new - s97"""
    return ['s_mov_b32 s97, s92', 's_memtime s[92:93]', 's_waitcnt lgkmcnt(0)', 's_sub_u32 s98, s92, s97', 's_add_u32 s%d, s%d, s98' % (acc, acc)]


# The schedule of one iteration: what is issued behind MFMA k (k = 0..127).
SCHED = dict(
    rd1_first=1, rd1_pattern=(0, 1),     # second sub-step's fragments: two reads in every three gaps from gap rd1_first on
    wait_m=31, bar_m=32,                 # s_waitcnt vmcnt(..) lgkmcnt(0) behind MFMA 31, s_barrier behind MFMA 32
    dma_w_first=34, dma_x_first=58, dma_step=3, dma_x_step=6,     # W(i+2) a piece every 3rd gap (one iteration to land), X(i+3) every 6th (two)
    rd0_first=67, rd0_pattern=(0, 1),
    rotate_at=104, loop_at=127, wait_end=126,
)


def body(dma_w, dma_x, vm_at_m, read_next, c0=False, loop_label=None, sched=SCHED):
    """One K-tile: 128 MFMAs with everything else between them.  dma_w / dma_x: None, 'cur' (this output tile's W(i+2) / X(i+3)),
    'next0' (the next output tile's first piece set of that operand: the descriptor's base is switched first) or 'next'."""
    gaps = {k: [] for k in range(-1, 128)}
    a, b, c, d, e = S_SLOT
    s = sched
    if s.get('no_dma'):
        dma_w = dma_x = None
        vm_at_m = None
    wait_m = 's_waitcnt lgkmcnt(0)' if vm_at_m is None else 's_waitcnt vmcnt(%d) lgkmcnt(0)' % vm_at_m
    # ---- in front of M: the second sub-step's fragments of tile i (slots a = X(i), b = W(i))
    gaps[0] += [addr('w', 1, b), addr('x', 1, a)]
    g = s['rd1_first']
    reads = [rd('w', 1, i) for i in range(8)] + [rd('x', 1, i) for i in range(8)]
    n = 0
    while n < 16:
        for o in s['rd1_pattern']:
            if n < 16:
                gaps[g + o].append(reads[n]); n += 1
        g += s.get('rd1_stride', 3)
    assert g - s.get('rd1_stride', 3) + max(s['rd1_pattern']) < s['wait_m']
    if s.get('trace'):
        gaps[s['wait_m'] - 1] += stamp_first()       # executes one MFMA in front of the wait; the wait retires it with everything else
        gaps[s['wait_m']] += [wait_m] + stamp_take(94) + ['s_mov_b32 s97, s92']
        gaps[s['bar_m']] += ['s_barrier', 's_memtime s[92:93]', 's_waitcnt lgkmcnt(0)', 's_sub_u32 s98, s92, s97', 's_add_u32 s95, s95, s98']
    else:
        gaps[s['wait_m']].append(wait_m)
        gaps[s['bar_m']].append('s_barrier')
    # ---- behind M: W(i+2) -> slot a, X(i+3) -> slot b (or the next output tile's pieces)
    for what, op, slot, first in ((dma_w, 'w', a, s['dma_w_first']), (dma_x, 'x', b, s['dma_x_first'])):
        if not what:
            continue
        step = s.get('dma_%s_step' % op, s['dma_step'])
        gaps[first - 1] += (srd_base(op, nxt=True) if what == 'next0' else []) + [dst(slot), m0(0)]
        for p in range(8):
            gaps[first + step * p].append(piece(op, p))
            if p < 7 and m0(p + 1):
                gaps[first + step * p + 1].append(m0(p + 1))
        gaps[first + step * 7 + 1] += advance(op)
    # ---- first sub-step's fragments of tile i + 1 (slots c = X(i+1), d = W(i+1)); their registers are free behind MFMA 63
    if read_next:
        g = s['rd0_first']
        assert g > 64
        gaps[g - 1] += [addr('w', 0, d), addr('x', 0, c)]
        reads = [rd('w', 0, i) for i in range(8)] + [rd('x', 0, i) for i in range(8)]
        n = 0
        while n < 16:
            for o in s['rd0_pattern']:
                if n < 16:
                    gaps[g + o].append(reads[n]); n += 1
            g += s.get('rd0_stride', 3)
        assert g < s['rotate_at']
        if s.get('trace'):
            gaps[s['wait_end'] - 1] += stamp_first()
            gaps[s['wait_end']] += ['s_waitcnt lgkmcnt(0)'] + stamp_take(96)
        else:
            gaps[s['wait_end']].append('s_waitcnt lgkmcnt(0)')
    gaps[s['rotate_at']] += rotate()
    if loop_label:
        gaps[s['loop_at']] += ['s_sub_u32 s%d, s%d, 1' % (S_CNT, S_CNT), 's_cmp_lg_u32 s%d, 0' % S_CNT]
    out = []
    for k in range(128):
        out.append(mfma(k, c0))
        for ins in gaps[k]:
            if s.get('no_reads') and ins.startswith('ds_read'):
                continue
            if s.get('no_barrier') and ins == 's_barrier':
                continue
            out.append(ins)
    if loop_label:
        out.append('s_cbranch_scc1 %s' % loop_label)
    return out


def setup():
    """descriptor flags, piece offsets, the ring registers from `ring`, loop count"""
    out = ['s_mov_b32 s%d, 0x80000000' % (SRD['x'] + 2), 's_mov_b32 s%d, 0x00020000' % (SRD['x'] + 3),
           's_mov_b32 s%d, 0x80000000' % (SRD['w'] + 2), 's_mov_b32 s%d, 0x00020000' % (SRD['w'] + 3)]
    ox, ow = S_OFF['x'], S_OFF['w']
    out.append('s_lshl_b32 s%d, %%[rowb], 3' % ox)                       # X piece p: rows 8 p
    for p in range(2, 8):
        out.append('s_add_u32 s%d, s%d, s%d' % (ox + p - 1, ox + p - 2, ox))
    out.append('s_mov_b32 s%d, %%[wpo]' % ow)                            # W piece p: (p & 1) wpo + (p >> 1) rows
    out.append('s_mov_b32 s%d, %%[rowb]' % (ow + 1))
    for p in range(3, 8):
        out.append('s_add_u32 s%d, s%d, %%[rowb]' % (ow + p - 1, ow + p - 3))
    if IOFF:                                                              # every piece's source offset += 3 072 - 1 024 (p & 3)
        for op in ('x', 'w'):
            out.append('s_mov_b32 s%d, 3072' % S_OFF0[op])
            for p in range(1, 8):
                out.append('s_add_u32 s%d, s%d, %d' % (S_OFF[op] + p - 1, S_OFF[op] + p - 1, 3072 - 1024 * (p & 3)))
    out.append('s_add_u32 s%d, %%[lds0], %d' % (S_END, 5 * 32768))
    out.append('s_add_u32 s%d, %%[lds0], %%[ring]' % S_SLOT[0])
    for i in range(1, 5):                                                 # next slot = + 32 KB, wrapping at the end of the ring
        sl, pr = S_SLOT[i], S_SLOT[i - 1]
        out += ['s_add_u32 s%d, s%d, 32768' % (sl, pr), 's_sub_u32 s%d, s%d, %d' % (S_TMP, sl, 5 * 32768), 's_cmp_ge_u32 s%d, s%d' % (sl, S_END),
                's_cselect_b32 s%d, s%d, s%d' % (sl, S_TMP, sl)]
    out.append('s_mov_b32 s%d, %%[wdst]' % S_WDST)
    out.append('s_sub_u32 s%d, %%[np], 4' % S_CNT)
    return out


def program(sched=SCHED, st='h'):
    a, b, c, d, e = S_SLOT
    S = ST[st]
    n_entry, n_first = min(63, 24 + S), min(63, 8 + S)
    out = []
    if sched.get('trace'):
        out += ['s_mov_b32 s94, 0', 's_mov_b32 s95, 0', 's_mov_b32 s96, 0', 's_mov_b32 s99, 0',
                's_memtime s[92:93]', 's_waitcnt lgkmcnt(0)', 's_mov_b32 s101, s92'] + stamp_first() + ['s_waitcnt lgkmcnt(0)'] + stamp_take(99)
    out += setup()
    out += srd_base('x') + srd_base('w')
    out += ['s_cmp_eq_u32 %[first], 0', 's_cbranch_scc1 .Lw4_pref_%=']
    # ---- nothing prefetched (a workgroup's first tile): X(0) -> a, W(0) -> b, X(1) -> c, W(1) -> d, X(2) -> e
    for op, slot in (('x', a), ('w', b), ('x', c), ('w', d), ('x', e)):
        out += [dst(slot)]
        for p in range(8):
            out += ([m0(p), 's_nop 0'] if m0(p) else []) + [piece(op, p)]
        out += advance(op)
    out += ['s_waitcnt vmcnt(8)', 's_branch .Lw4_go_%=']           # (X(1), W(1) too: the first M's allowance is the prefetched case's)
    # ---- the previous block has issued them: the descriptors continue behind them
    out += ['.Lw4_pref_%=:'] + advance('x', 3) + advance('w', 2) + ['s_waitcnt vmcnt(%d)' % n_entry]
    out += ['.Lw4_go_%=:', 's_barrier']
    out += [addr('w', 0, b), addr('x', 0, a)]
    out += [rd('w', 0, i) for i in range(8)] + [rd('x', 0, i) for i in range(8)]
    out += ['s_waitcnt lgkmcnt(0)']
    if sched.get('trace'):
        out += ['s_memtime s[92:93]', 's_waitcnt lgkmcnt(0)', 's_mov_b32 s100, s92']
    out += body('cur', 'cur', n_first, True, c0=True, sched=sched)                 # i = 0
    out += ['s_cmp_eq_u32 s%d, 0' % S_CNT, 's_cbranch_scc1 .Lw4_tail_%=', '.p2align 4', '.Lw4_loop_%=:']
    out += body('cur', 'cur', 8, True, loop_label='.Lw4_loop_%=', sched=sched)     # i = 1 .. np - 4
    out += ['.Lw4_tail_%=:']
    out += body('cur', 'next0', 8, True, sched=sched)        # i = np - 3: W(np-1), X'(0)
    out += body('next0', 'next', 8, True, sched=sched)       # i = np - 2: W'(0), X'(1)
    out += body('next', 'next', None, False, sched=sched)    # i = np - 1: W'(1), X'(2); M = every wave is done with the last slots (no data awaited)
    out += ['s_sub_u32 %%[ring], s%d, %%[lds0]' % a]
    out += ['s_nop 15', 's_nop 15']           # the last MFMAs have written their accumulators before anything reads them
    if sched.get('trace'):
        out += ['s_memtime s[92:93]', 's_waitcnt lgkmcnt(0)', 's_sub_u32 %[t_pro], s100, s101', 's_sub_u32 %[t_loop], s92, s100',
                's_mov_b32 %[t_wait], s94', 's_mov_b32 %[t_bar], s95', 's_mov_b32 %[t_end], s96', 's_mov_b32 %[t_cal], s99']
    return out


def clobbers():
    c = ['"memory"', '"scc"', '"m0"']
    c += ['"v%d"' % i for i in range(120, 256)]
    c += ['"a%d"' % i for i in range(256)]
    c += ['"s%d"' % i for i in range(58, 102)]
    return c


# Development variants (compiled under VG_DEV only; VG_GEMM_W4 = 1 + index): ablations that give WRONG results (what does the loop cost
# without its DMA / reads / barrier?) and alternative schedules.
VARIANTS = [
    dict(),                                  # 0: the product schedule
    dict(no_dma=True),                       # 1
    dict(no_reads=True),                     # 2
    dict(no_barrier=True),                   # 3
    dict(no_dma=True, no_reads=True, no_barrier=True),   # 4: MFMAs only
    dict(dma_step=2, dma_x_first=50),                    # 5: the first schedule (a piece every second gap)
    dict(dma_step=4, dma_x_first=66),                    # 6
    dict(trace=True),                                    # 7: the product schedule with cycle stamps (vg_gemm_trace var 50)
    dict(dma_x_step=3, rotate_at=100),                   # 8: the schedule until the end of round 6's first pass: X pieces every 3rd gap too
    dict(dma_x_step=8, dma_x_first=60, rotate_at=120, rd0_first=67),   # 9: ... every eighth
    dict(dma_x_step=3, rd0_first=97, rotate_at=124),     # 10: next tile's first fragments read late (behind the pieces)
    dict(dma_x_first=92, dma_x_step=4, rotate_at=122),   # 11: X pieces behind the next tile's first fragments
    dict(dma_w_step=4, dma_x_first=64, dma_x_step=6, rotate_at=110),          # 12
    dict(dma_x_step=5, rotate_at=104),                                        # 13
    dict(dma_x_step=7, dma_x_first=57, rotate_at=110),                        # 14
    dict(rd1_pattern=(0,), rd1_stride=2, wait_m=39, bar_m=40, dma_w_first=42, dma_x_first=66, rd0_first=68, rotate_at=112),   # 15: a read every 2nd gap, M at 40
    dict(dma_w_step=2, dma_x_first=52),                                       # 16: W pieces every 2nd gap
    dict(dma_x_first=62, rotate_at=108),                                      # 17
    dict(rd0_pattern=(0,), rd0_stride=2, rd0_first=67, rotate_at=104),        # 18: next tile's first fragments one read per 2 gaps (67..97)
    dict(rd1_pattern=(0,), rd1_stride=1, rd1_first=1),                        # 19: second sub-step's fragments one read per gap (1..16)
]


def render():
    lines = ['// GENERATED by gen_gemm_w4.py -- do not edit; see that file for what the schedule is and why.']
    for vi, var in enumerate(VARIANTS):
        sched = dict(SCHED); sched.update(var)
        if vi == 1:
            lines.append('#ifdef VG_DEV')
        for st in ('h', 'f') if vi == 0 else ('h',):
            lines.append('#define VG_W4_ASM_%d%s \\' % (vi, st.upper()))
            for ins in program(sched, st):
                lines.append('    "%s\\n" \\' % ins)
            lines.append('    ""')
    lines.append('#define VG_W4_DEV_RUNS \\')
    for vi in range(1, len(VARIANTS)):
        lines.append('    else if constexpr (VAR == %d) { VG_W4_RUN%s(%dH); } \\' % (vi, '_TRACE' if VARIANTS[vi].get('trace') else '', vi))
    lines.append('')
    lines.append('#define VG_W4_DEV_CASES(EPI, LN) \\')
    for vi in range(1, len(VARIANTS)):
        if not VARIANTS[vi].get('trace'):
            lines.append('    case %d: return launch_gemm_w4<EPI, LN, %d>(X, Wt, bias, C, resid, M, N, K, ldc, st); \\' % (vi + 1, vi))
    lines.append('')
    lines.append('#endif  // VG_DEV')
    lines.append('#define VG_W4_NVAR %d' % len(VARIANTS))
    lines.append('#define VG_W4_TRACE_VAR %d' % [i for i, v in enumerate(VARIANTS) if v.get('trace')][0])
    lines.append('#define VG_W4_STORES_H %d' % ST['h'])
    lines.append('#define VG_W4_STORES_F %d' % ST['f'])
    cl = clobbers()
    lines.append('#define VG_W4_CLOBBERS \\')
    for i in range(0, len(cl), 16):
        lines.append('    ' + ', '.join(cl[i:i + 16]) + (', \\' if i + 16 < len(cl) else ''))
    return '\n'.join(lines) + '\n'


if __name__ == '__main__':
    text = render()
    if '--check' in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == text
        print('gemm_w4_loop.inc', 'up to date' if ok else 'STALE')
        sys.exit(0 if ok else 1)
    with open(OUT, 'w') as f:
        f.write(text)
    print('wrote', OUT, len(program()), 'instructions per block')
