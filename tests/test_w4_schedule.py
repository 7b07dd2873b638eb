"""Static check of the hand-scheduled K loop of k_gemm_f16_w4 (vilgod_amd/csrc/gen_gemm_w4.py): one wave's instruction stream is
interpreted symbolically -- scalar registers with concrete values, LDS slots and fragment registers with the (operand, K-tile, half,
fragment) they hold -- and every protocol rule the block relies on is checked at every instruction:

  * an MFMA of K-tile i, sub-step h multiplies X fragment (i, h, mi) by W fragment (i, h, ni), each read by a ds_read the wave has
    waited for (lgkmcnt), into accumulator tile 8 ni + mi -- every tile exactly once per sub-step, the first sub-step of a block with C = 0;
  * a ds_read takes its fragment from a slot that holds the K-tile it wants, whose DMA pieces this wave has retired by a counted vmcnt
    BEFORE a barrier the wave has passed since (the pieces of the other waves: same program, same counts);
  * a DMA piece is written into a slot only after every read of the slot's previous content was waited for and a barrier passed;
  * a piece's source is the K-tile the ring expects there (this tile's, or the NEXT output tile's first five slots in the last three
    iterations), its LDS rows are this wave's (M0 + instruction offset), and the ring is left the way the next block assumes;
  * the vmcnt allowances: with S younger vector-memory operations of the epilogue in flight (prefetched entry) the waits still cover
    exactly the pieces they must.

"Place reads by the vmcnt / barrier count, never by clean runs" (cdna_hip_programming.md): a race here returns stale LDS bytes
without any fault, and only when the DMA happens to be late."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('gen_gemm_w4', os.path.join(ROOT, 'vilgod_amd', 'csrc', 'gen_gemm_w4.py'))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)

SLOT = 32768
ROWB = 1536                      # K = 768 halves
XBASE, WBASE, NXBASE, NWBASE = 0x10_0000_0000, 0x20_0000_0000, 0x30_0000_0000, 0x40_0000_0000
WPO = 64 * ROWB


class Sim:
    def __init__(self, prog, np_, first, ring, st_kind, stores_in_flight):
        self.prog, self.np, self.first = prog, np_, first
        self.lds0 = 4096                 # any LDS base
        self.ops = {'xlo': XBASE & 0xFFFFFFFF, 'xhi': XBASE >> 32, 'wlo': WBASE & 0xFFFFFFFF, 'whi': WBASE >> 32,
                    'nxlo': NXBASE & 0xFFFFFFFF, 'nxhi': NXBASE >> 32, 'nwlo': NWBASE & 0xFFFFFFFF, 'nwhi': NWBASE >> 32,
                    'rowb': ROWB, 'wpo': WPO, 'lds0': self.lds0, 'wdst': 2 * 8192, 'np': np_, 'first': first, 'ring': ring * SLOT}
        self.s = {}                      # scalar registers
        self.scc = 0
        self.m0 = None
        self.v = {}                      # address VGPRs v120..v123 -> (slot base value, operand key 'xo0' ...)
        self.frag = {}                   # first register of a fragment -> ('x'|'w', tile, half, idx, ready)
        self.pending_reads = []          # fragments whose ds_read is in flight (lgkmcnt)
        # slots: physical index -> dict(content=(op, tile, gen), pieces=set, retired=bool, visible=bool, reads_open=bool, reads_waited=bool, free=bool)
        self.slots = [dict(content=None, pieces=0, retired=False, visible=False, reads_unwaited=False, reads_since_barrier=False) for _ in range(5)]
        self.vm = []                     # outstanding vector-memory ops, oldest first: ('piece', slot index) | ('store',)
        self.epoch = 0
        self.acc_written = {}
        self.mfma_count = 0
        self.iter_tiles = []             # MFMA log: (tile, half, ni, mi)
        if not first:
            # what the previous block and its epilogue left behind: X'(0) W'(0) X'(1) W'(1) X'(2) in ring order, all five in flight behind
            # nothing, then S stores
            for q, (op, t) in enumerate((('x', 0), ('w', 0), ('x', 1), ('w', 1), ('x', 2))):
                sl = (ring + q) % 5
                self.slots[sl] = dict(content=(op, t), pieces=8, retired=False, visible=False, reads_unwaited=False, reads_since_barrier=False)
                self.vm += [('piece', sl)] * 8
            self.vm += [('store',)] * stores_in_flight
            self.retire_to(63)               # (the 6-bit counter: the epilogue's later stores could only issue as the oldest pieces retired)
        self.expected_x_tile = 3 if not first else 0     # next cur-tile index the X descriptor must deliver
        self.expected_w_tile = 2 if not first else 0

    # ---- helpers
    def val(self, tok):
        tok = tok.strip().rstrip(',')
        m = re.fullmatch(r'%\[(\w+)\]', tok)
        if m:
            return self.ops[m.group(1)]
        if re.fullmatch(r's\d+', tok):
            return self.s[int(tok[1:])]
        if tok == 'm0':
            return self.m0
        return int(tok, 0)

    def setreg(self, tok, value):
        tok = tok.strip().rstrip(',')
        value &= 0xFFFFFFFF
        m = re.fullmatch(r'%\[(\w+)\]', tok)
        if m:
            self.ops[m.group(1)] = value
        elif tok == 'm0':
            self.m0 = value
        else:
            self.s[int(tok[1:])] = value

    def slot_of(self, lds_addr):
        off = lds_addr - self.lds0
        assert 0 <= off < 5 * SLOT, f'LDS address {lds_addr} outside the ring'
        return off // SLOT, off % SLOT

    def run(self):
        labels = {ins[:-1]: i for i, ins in enumerate(self.prog) if ins.endswith(':')}
        pc = 0
        steps = 0
        while pc < len(self.prog):
            ins = self.prog[pc]
            steps += 1
            assert steps < 200_000
            nxt = pc + 1
            op, _, rest = ins.partition(' ')
            args = [a.strip() for a in rest.split(',')] if rest else []
            if ins.endswith(':') or op in ('.p2align', 's_nop'):
                pass
            elif op == 's_mov_b32':
                self.setreg(args[0], self.val(args[1]))
            elif op in ('s_add_u32', 's_addc_u32'):
                r = self.val(args[1]) + self.val(args[2]) + (self.scc if op == 's_addc_u32' else 0)
                self.scc = 1 if r > 0xFFFFFFFF else 0
                self.setreg(args[0], r)
            elif op in ('s_sub_u32', 's_subb_u32'):
                r = self.val(args[1]) - self.val(args[2]) - (self.scc if op == 's_subb_u32' else 0)
                self.scc = 1 if r < 0 else 0
                self.setreg(args[0], r)
            elif op == 's_lshl_b32':
                self.setreg(args[0], self.val(args[1]) << self.val(args[2]))
            elif op == 's_cmp_eq_u32':
                self.scc = int(self.val(args[0]) == self.val(args[1]))
            elif op == 's_cmp_lg_u32':
                self.scc = int(self.val(args[0]) != self.val(args[1]))
            elif op == 's_cmp_ge_u32':
                self.scc = int(self.val(args[0]) >= self.val(args[1]))
            elif op == 's_cselect_b32':
                self.setreg(args[0], self.val(args[1]) if self.scc else self.val(args[2]))
            elif op == 's_cbranch_scc1':
                if self.scc:
                    nxt = labels[args[0]]
            elif op == 's_branch':
                nxt = labels[args[0]]
            elif op == 's_barrier':
                self.barrier()
            elif op == 's_waitcnt':
                self.waitcnt(rest)
            elif op == 'v_add_u32':
                m = re.fullmatch(r'%\[(\w+)\]', args[2])
                self.v[int(args[0][1:])] = (self.val(args[1]), m.group(1))
            elif op == 'ds_read_b128':
                self.ds_read(args)
            elif op == 'buffer_load_dwordx4':
                self.piece(rest)
            elif op == 'v_mfma_f32_16x16x32_f16':
                self.mfma(rest)
            else:
                raise AssertionError(f'instruction the checker does not know: {ins}')
            pc = nxt
        return self

    # ---- memory protocol
    def barrier(self):
        self.epoch += 1
        for sl in self.slots:
            if sl['retired']:
                sl['visible'] = True             # every wave retired its pieces before it arrived here (same program, same counts)
            if not sl['reads_unwaited']:
                sl['reads_since_barrier'] = False

    def retire_to(self, keep):
        while len(self.vm) > keep:
            o = self.vm.pop(0)                   # in-order retirement
            if o[0] == 'piece':
                sl = self.slots[o[1]]
                sl['pieces'] -= 1
                if sl['pieces'] == 0:
                    sl['retired'] = True

    def waitcnt(self, rest):
        m = re.search(r'vmcnt\((\d+)\)', rest)
        if m:
            self.retire_to(int(m.group(1)))
        if 'lgkmcnt(0)' in rest:
            for key in self.pending_reads:
                self.frag[key] = self.frag[key][:4] + (True,)
            self.pending_reads = []
            for sl in self.slots:
                sl['reads_unwaited'] = False

    def ds_read(self, args):
        dst = int(re.match(r'v\[(\d+):', args[0]).group(1))
        toks = args[1].split()
        base, key = self.v[int(toks[0][1:])]
        off = int(toks[1].split(':')[1]) if len(toks) > 1 else 0
        sl_i, _ = self.slot_of(base)
        assert (base - self.lds0) % SLOT == 0
        sl = self.slots[sl_i]
        opn, half = key[0], int(key[2])
        assert sl['content'] is not None and sl['content'][0] == opn, f'read of {key} from slot {sl_i} holding {sl["content"]}'
        assert sl['retired'] and sl['visible'], f'fragment read from slot {sl_i} ({sl["content"]}) before its pieces were retired and a barrier passed'
        assert off % 2048 == 0 and 0 <= off // 2048 < 8
        self.frag[dst] = (opn, sl['content'][1], half, off // 2048, False)
        self.pending_reads.append(dst)
        sl['reads_unwaited'] = True
        sl['reads_since_barrier'] = True

    def piece(self, rest):
        m = re.match(r'%\[(d[vw])(\d)\], s\[(\d+):\d+\], (\S+) offen(?: offset:(\d+))? lds', rest)
        assert m, rest
        kind, odd, srd, so, ioff = m.group(1), int(m.group(2)), int(m.group(3)), m.group(4), int(m.group(5) or 0)
        opn = 'x' if kind == 'dv' else 'w'
        assert srd == gen.SRD[opn]
        dest = self.m0 + ioff                                   # LDS: M0 + instruction offset (+ 16 B per lane)
        sl_i, within = self.slot_of(dest)
        p = (within - self.ops['wdst']) // 1024
        assert within == self.ops['wdst'] + 1024 * p and 0 <= p < 8, 'a piece outside this wave\'s rows of the slot'
        assert odd == (p & 1)
        src = (self.s[srd] | (self.s[srd + 1] << 32)) + self.val(so) + ioff          # + the per-lane offset
        rows = 8 * p * ROWB if opn == 'x' else (p & 1) * WPO + (p >> 1) * ROWB
        for base, nxt in ((XBASE if opn == 'x' else WBASE, False), (NXBASE if opn == 'x' else NWBASE, True)):
            d = src - base - rows
            if 0 <= d < 128 * 4096 and d % 128 == 0:
                tile, is_next = d // 128, nxt
                break
        else:
            raise AssertionError(f'piece source {src:#x} is no K-tile of either output tile')
        sl = self.slots[sl_i]
        if p == 0:
            assert not sl['reads_since_barrier'] and not sl['reads_unwaited'], f'slot {sl_i} refilled before its readers were waited for and a barrier passed'
            assert sl['pieces'] == 0, f'slot {sl_i} refilled while its previous pieces are in flight'
            self.slots[sl_i] = sl = dict(content=(opn, ('n', tile) if is_next else tile), pieces=0, retired=False, visible=False,
                                         reads_unwaited=False, reads_since_barrier=False)
        assert sl['content'] == (opn, ('n', tile) if is_next else tile), 'the eight pieces of a slot belong to one K-tile'
        sl['pieces'] += 1
        self.retire_to(62)                       # a wave with 63 operations outstanding issues the next one when the oldest has retired
        self.vm.append(('piece', sl_i))

    def mfma(self, rest):
        m = re.match(r'a\[(\d+):\d+\], v\[(\d+):\d+\], v\[(\d+):\d+\], (\S+)', rest)
        acc, fa, fb, c = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4)
        x, w = self.frag[fa], self.frag[fb]
        assert x[0] == 'x' and w[0] == 'w' and x[4] and w[4], 'an MFMA on a fragment that was not read or not waited for'
        assert x[1] == w[1] and x[2] == w[2], f'X fragment of K-tile {x[1]} half {x[2]} against W fragment of {w[1]} half {w[2]}'
        assert acc == 4 * (w[3] * 8 + x[3]), 'accumulator tile 8 ni + mi'
        if c == '0':
            assert x[1] == 0 and x[2] == 0, 'C = 0 only in the first sub-step of a block'
        else:
            assert c == f'a[{acc}:{acc + 3}]' and not (x[1] == 0 and x[2] == 0)
        self.iter_tiles.append((x[1], x[2], w[3], x[3]))


def simulate(np_, first, ring, st):
    prog = [i for i in gen.program(dict(gen.SCHED), st)]
    S = gen.ST[st]
    return Sim(prog, np_, first, ring, st, 0 if first else S).run()


@pytest.mark.parametrize('st', ['h', 'f'])
@pytest.mark.parametrize('first', [1, 0])
@pytest.mark.parametrize('np_', [4, 5, 7, 12, 48])
def test_w4_block_protocol(np_, first, st):
    for ring in (0, 3):
        sim = simulate(np_, first, ring, st)
        # every (K-tile, sub-step) multiplied every one of the 64 accumulator tiles exactly once, K-tiles and sub-steps in order
        assert len(sim.iter_tiles) == np_ * 128
        for i in range(np_):
            for h in range(2):
                chunk = sim.iter_tiles[(2 * i + h) * 64:(2 * i + h + 1) * 64]
                assert all(t == i and hh == h for t, hh, _, _ in chunk)
                assert sorted((ni, mi) for _, _, ni, mi in chunk) == [(a, b) for a in range(8) for b in range(8)]
        # what the block leaves: the next tile's first five slots in ring order, in flight or landed, nothing else outstanding
        ring_out = sim.ops['ring'] // SLOT
        want = [('x', ('n', 0)), ('w', ('n', 0)), ('x', ('n', 1)), ('w', ('n', 1)), ('x', ('n', 2))]
        assert [sim.slots[(ring_out + q) % 5]['content'] for q in range(5)] == want
        outstanding = [o for o in sim.vm if o[0] == 'piece']
        assert len(outstanding) <= 40 and not sim.pending_reads
        assert [sim.slots[o[1]]['content'] for o in outstanding[::8]] == want[5 - len(outstanding) // 8:]


def test_w4_allowance_is_tight_enough():
    """with FEWER younger operations in flight than the generator assumes (a compiler that merged stores) the prefetched entry's waits
    would let a read run ahead of its piece: the checker must see that (it is what tests/test_abi.py guards against in the binary)."""
    prog = [i for i in gen.program(dict(gen.SCHED), 'h')]
    with pytest.raises(AssertionError, match='before its pieces were retired'):
        Sim(prog, 12, 0, 0, 'h', gen.ST['h'] - 9).run()
