#!/usr/bin/env python3
"""Generate neuralplane_amd/csrc/np_mlp_asm.inc — the hand-scheduled gfx950 bodies of the aero MLPs.

One `asm volatile` statement evaluates ALL nets of a class (same shape, same inputs, KBLOB records
back to back): weights stream through the scalar unit in 16-dword chunks, double-buffered in groups
of two chunks (4 x 16 fixed SGPRs), each group's `s_load_dwordx16` pair issued one group (32 FMAs)
ahead of its single `s_waitcnt lgkmcnt(0)` — also across net boundaries, the records being
contiguous.  FMAs are `v_pk_fma_f32 acc[2], s[2], x(broadcast)`: two neurons of a layer per
instruction, the weight pair an SGPR operand (130 TFLOP/s issue ceiling on MI355X,
tools/microbench/fma_rates.hip).  Every accumulator keeps the spec's order
`acc = bias; acc = fma(W[j][k], x[k], acc)`, k ascending, so results are bit-identical to the C++
path and to the oracle.

Why asm: hipcc cannot be steered into this schedule (it either spills SGPRs lane-by-lane into VGPRs
or sinks every load next to its use), and asm loads whose destinations the compiler can see are
unsafe (it spills / re-uses SGPRs of in-flight loads; tools/audit_smem_asm.py).  Inside ONE
statement nothing is visible to the compiler: temporaries are fixed registers declared as clobbers.

Record layout of one net (floats) — must match np_nets.h::asm_record_len / pack_kblob:
  per hidden layer (in -> out):  bias[out] (+pad to even), then for k < in: W[.][k] as a row of
                                 `out` weights (+pad to even)            [pairs are even-aligned]
  final layer (in -> 1):         bias, W[0][0..in) (+pad to even)
  out_std, out_mean, then zero padding to a multiple of 32 floats (whole groups)
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.environ.get('NPF16_GEN_OUTDIR') or os.path.join(os.path.dirname(HERE), 'neuralplane_amd', 'csrc')   # tests regenerate into a temp dir
OUT = os.path.join(CSRC, 'np_mlp_asm.inc')

SHAPES = [(1, 20, 10, 0), (3, 20, 10, 0), (2, 20, 10, 0), (2, 20, 10, 5), (2, 20, 20, 10), (1, 20, 10, 5), (2, 20, 10, 10)]

# fixed registers (declared as clobbers of every statement)
CPG = 3          # 16-dword chunks per group (a group is retired by ONE s_waitcnt lgkmcnt(0))
NBUF = 2 * CPG   # double-buffered groups
GROUP = 16 * CPG # floats per group
S_W0 = 4         # s[4:99] six 16-dword weight buffers
S_BASE = 100     # s[100:101] record pointer
S_CNT = 'vcc_lo' # remaining nets
S_NEXT = 2       # s[2:3] (phase functions only): pointer of the record that follows the current one in the SEQUENCE
V_H = [70, 90, 110]   # hidden activations of layer 1/2/3 (20, 20, 10 registers)
V_X = 120        # v120,v122,v124: the (normalised) inputs, even registers
V_Y = 126        # v[126:127] the two partial sums of the output layer; v126 = net output
V_ADDR = 121     # LDS byte address of the output slot (the odd register between the input pairs: never read as an input)
HALF_LOADS = os.environ.get('NPF16_GEN_HALF_LOADS') == '1'   # TIMING EXPERIMENT ONLY (wrong results): every other group's loads are not issued — what halving the scalar weight traffic would gain
GEN_DUP = os.environ.get('NPF16_GEN_DUP') == '1'   # TIMING EXPERIMENT ONLY: every VALU instruction of a net body is issued twice (second copy on
                                                    # registers + 58), i.e. two accumulator sets per weight load — what 'two aircraft per lane' would cost
V_CLOBBER = list(range(68 if os.environ.get('NPF16_GEN_EXTRA_MOVS') == '1' else 70, 186 if GEN_DUP else 128))
S_CLOBBER = list(range(S_W0, S_W0 + 16 * NBUF)) + [S_BASE, S_BASE + 1]


def pad2(n):
    return n + (n & 1)


def record_len(IN, H1, H2, H3):
    hid = [H1, H2] + ([H3] if H3 else [])
    n, prev = 0, IN
    for h in hid:
        n += pad2(h) + prev * pad2(h)
        prev = h
    n += 2 + pad2(prev) + 2    # final layer: (bias, 0) pair, W[0][0..in) padded to even; then out_std, out_mean
    return (n + GROUP - 1) // GROUP * GROUP


class Body:
    """Instruction list of one net evaluation, given the buffer parity of its first group."""

    def __init__(self, shape, parity, use_next=False):
        self.use_next = use_next  # overrun prefetches (first group of the FOLLOWING record) go through S_NEXT
        self.IN, self.H1, self.H2, self.H3 = shape
        self.len = record_len(*shape)
        self.nch = self.len // 16
        self.parity = parity  # 0: first group in the first CPG buffers ; 1: in the other CPG buffers
        self.ins = []
        self.pos = 0
        self.cur_group = -1
        # for the dual bodies (second_set): instructions that set B does not repeat (bias moves) and instructions whose set-B
        # version is special and goes FIRST (the first FMA of a chain takes the bias from set A's freshly initialised register)
        self.advanced = False
        self.skip_dup = set()
        self.b_first = {}

    def sreg(self, pos):
        c, l = divmod(pos, 16)
        b = (c + CPG * self.parity) % NBUF
        return S_W0 + 16 * b + l

    def buf_of_chunk(self, c):
        return S_W0 + 16 * ((c + CPG * self.parity) % NBUF)

    def touch(self, pos):
        """Called before an instruction that reads record position `pos`: emits the group boundary
        (wait for this group, issue the loads of the next one) when a new group starts."""
        g = pos // GROUP
        while self.cur_group < g:
            self.cur_group += 1
            c = CPG * self.cur_group
            self.ins.append('s_waitcnt lgkmcnt(0)')
            skip = HALF_LOADS and self.cur_group % 2 == 1
            for cc in range(c + CPG, c + 2 * CPG):  # next group (runs into the next record when cc >= nch)
                if self.use_next == 'advance' and cc >= self.nch:
                    # two-set phases: the record pointer itself moves on (by the byte distance the caller left in vcc_hi) as soon as
                    # the last group of this record has been requested — no second pointer pair, s2 / s3 stay free for the compiler
                    if not self.advanced:
                        self.ins.append(f's_add_u32 s{S_BASE}, s{S_BASE}, vcc_hi')
                        self.ins.append(f's_addc_u32 s{S_BASE + 1}, s{S_BASE + 1}, 0')
                        self.advanced = True
                    base, off = S_BASE, (cc - self.nch) * 64
                elif self.use_next and cc >= self.nch:
                    base, off = S_NEXT, (cc - self.nch) * 64
                else:
                    base, off = S_BASE, cc * 64
                if not skip:
                    self.ins.append(f's_load_dwordx16 s[{self.buf_of_chunk(cc)}:{self.buf_of_chunk(cc) + 15}], '
                                    f's[{base}:{base + 1}], 0x{off:x}')

    def spair(self, pos):
        assert pos % 2 == 0
        self.touch(pos)
        r = self.sreg(pos)
        assert r % 2 == 0
        return f's[{r}:{r + 1}]'

    def s1(self, pos):
        self.touch(pos)
        return f's{self.sreg(pos)}'

    @staticmethod
    def vpair(r):
        assert r % 2 == 0
        return f'v[{r}:{r + 1}]'

    def dense(self, n_in, n_out, in_regs, out_base):
        """hidden layer with ReLU: bias row then n_in weight rows, pairs via v_pk_fma_f32."""
        row = pad2(n_out)
        p0 = self.pos
        no_bias = os.environ.get('NPF16_GEN_NO_BIAS') == '1'   # TIMING EXPERIMENT ONLY (wrong results): upper bound of what
        if not no_bias:                                        # getting the bias moves off the VALU could gain
            for j in range(0, n_out - 1, 2):
                sp = self.spair(p0 + j)
                self.skip_dup.add(len(self.ins))
                self.ins.append(f'v_pk_mov_b32 {self.vpair(out_base + j)}, {sp}, {sp} op_sel:[0,1]')
                if os.environ.get('NPF16_GEN_EXTRA_MOVS') == '1':   # TIMING EXPERIMENT ONLY: what one bias move costs
                    self.ins.append(f'v_pk_mov_b32 v[68:69], {sp}, {sp} op_sel:[0,1]')
            if n_out & 1:
                s_b = self.s1(p0 + n_out - 1)
                self.skip_dup.add(len(self.ins))
                self.ins.append(f'v_mov_b32 v{out_base + n_out - 1}, {s_b}')
        for k in range(n_in):
            pk = p0 + row * (k + 1)
            xr = in_regs[k]
            e = xr & 1
            xp = self.vpair(xr - e)
            # ReLU = the clamp modifier of the LAST fused multiply-add of every accumulator: activations are carried divided
            # by 2^ACT_SHIFT (np_nets.h), so clamp's upper bound 1.0 is 2^40 in real terms; no v_max_f32 per neuron
            relu = ' clamp' if k == n_in - 1 else ''
            for j in range(0, n_out - 1, 2):
                sp = self.spair(pk + j)
                acc = self.vpair(out_base + j)
                if no_bias and k == 0:
                    self.ins.append(f'v_pk_mul_f32 {acc}, {sp}, {xp} op_sel:[0,{e}] op_sel_hi:[1,{e}]')
                    continue
                if k == 0 and not no_bias:   # set B: acc_B = fma(w, x_B, bias) with the bias read from set A's register
                    xpb, accb = self.vpair(xr - e + SET_B), self.vpair(out_base + j + SET_B)
                    self.b_first[len(self.ins)] = f'v_pk_fma_f32 {accb}, {sp}, {xpb}, {acc} op_sel:[0,{e},0] op_sel_hi:[1,{e},1]{relu}'
                self.ins.append(f'v_pk_fma_f32 {acc}, {sp}, {xp}, {acc} op_sel:[0,{e},0] op_sel_hi:[1,{e},1]{relu}')
            if n_out & 1:
                j = n_out - 1
                s_w = self.s1(pk + j)
                if k == 0 and not no_bias:
                    self.b_first[len(self.ins)] = f'v_fma_f32 v{out_base + j + SET_B}, {s_w}, v{xr + SET_B}, v{out_base + j}' + (' clamp' if relu else '')
                if relu:
                    self.ins.append(f'v_fma_f32 v{out_base + j}, {s_w}, v{xr}, v{out_base + j} clamp')
                else:
                    self.ins.append(f'v_fmac_f32 v{out_base + j}, {s_w}, v{xr}')
        self.pos = p0 + row * (n_in + 1)

    def final(self, n_in, in_regs):
        """output layer in -> 1 as TWO interleaved partial chains in one packed accumulator (numerics spec):
        acc = (bias, 0); acc = (fma(w[2i], x[2i], acc.lo), fma(w[2i+1], x[2i+1], acc.hi)) for i ascending; an odd last input goes to
        the low chain; y = acc.lo + acc.hi.  Half the instructions of the serial chain (5 + 1 instead of 10 for the 10-wide layers)."""
        p0 = self.pos
        assert in_regs[0] % 2 == 0 and all(in_regs[k] == in_regs[0] + k for k in range(n_in))
        sp = self.spair(p0)
        self.skip_dup.add(len(self.ins))
        self.ins.append(f'v_pk_mov_b32 {self.vpair(V_Y)}, {sp}, {sp} op_sel:[0,1]')
        for k in range(0, n_in - 1, 2):
            sw = self.spair(p0 + 2 + k)
            xp = self.vpair(in_regs[k])
            acc = self.vpair(V_Y)
            if k == 0:   # set B takes (bias, 0) from set A's freshly initialised pair
                self.b_first[len(self.ins)] = f'v_pk_fma_f32 {self.vpair(V_Y + SET_B)}, {sw}, {self.vpair(in_regs[k] + SET_B)}, {acc}'
            self.ins.append(f'v_pk_fma_f32 {acc}, {sw}, {xp}, {acc}')
        if n_in & 1:
            k = n_in - 1
            assert k > 0
            self.ins.append(f'v_fmac_f32 v{V_Y}, {self.s1(p0 + 2 + k)}, v{in_regs[k]}')
        self.ins.append(f'v_add_f32 v{V_Y}, v{V_Y}, v{V_Y + 1}')
        self.pos = p0 + 2 + pad2(n_in)
        # unnormalize: X * std + mean, two roundings (hifi_F16_AeroData.py:36-37)
        self.ins.append(f'v_mul_f32 v{V_Y}, {self.s1(self.pos)}, v{V_Y}')
        self.ins.append(f'v_add_f32 v{V_Y}, {self.s1(self.pos + 1)}, v{V_Y}')
        self.pos += 2

    def build(self):
        x_regs = [V_X, V_X + 2, V_X + 4][:self.IN]
        self.dense(self.IN, self.H1, x_regs, V_H[0])
        h1 = [V_H[0] + i for i in range(self.H1)]
        self.dense(self.H1, self.H2, h1, V_H[1])
        h2 = [V_H[1] + i for i in range(self.H2)]
        if self.H3:
            self.dense(self.H2, self.H3, h2, V_H[2])
            self.final(self.H3, [V_H[2] + i for i in range(self.H3)])
        else:
            self.final(self.H2, h2)
        assert self.pos <= self.len, (self.pos, self.len)
        # make sure every group of the record passed its boundary (padding groups included), so that
        # the stream position is exactly one record further when the next net starts
        self.touch(self.len - 1)
        assert self.cur_group == self.len // GROUP - 1
        if GEN_DUP:
            self.ins = second_set(self.ins)   # the experiment duplicates everything, bias moves included
        return self.ins


SET_B = 58   # the second accumulator set of the dual bodies lives SET_B registers above the first (v128-v185)


def second_set(ins_list, body=None):
    """every VALU instruction once more, on the registers of the second set (same SGPR weight operands) — except the bias
    moves: set B's first FMA of every chain reads the bias out of set A's register (before set A's own first FMA overwrites
    it), so acc_B = fma(w0, x0_B, bias) needs no move of its own.  The chains stay acc = bias; acc = fma(w_k, x_k, acc)."""
    import re
    skip = body.skip_dup if body is not None else set()
    first = body.b_first if body is not None else {}

    def shift(m):
        if m.group(1) is not None:
            a, b = int(m.group(1)), int(m.group(2))
            return f'v[{a + SET_B}:{b + SET_B}]' if a >= 70 else m.group(0)
        r = int(m.group(3))
        return f'v{r + SET_B}' if r >= 70 else m.group(0)
    out = []
    for i, ins in enumerate(ins_list):
        if i in first:
            out.append(first[i])
            out.append(ins)
            continue
        out.append(ins)
        if ins.startswith('v_') and i not in skip:
            out.append(re.sub(r'v\[(\d+):(\d+)\]|\bv(\d+)\b', shift, ins))
    return out


# ------------------------------------------------------------------------------------------------
# Two accumulator sets, "neuron-major" (the bodies of np_mlp_asm_dual.inc since round 2): a packed register holds ONE neuron
# of BOTH sets (lo = set A, hi = set B), every v_pk_fma_f32 broadcasts ONE weight — either half of an SGPR pair, chosen by
# op_sel — to both halves, and the FIRST FMA of every chain reads its bias from the other half of the same SGPR pair:
#     v_pk_fma_f32 acc, s[w:b], x, s[w:b] op_sel:[0,0,1] op_sel_hi:[0,1,1]      acc.lo = fma(w, xA, b), acc.hi = fma(w, xB, b)
# (one SGPR pair used twice is one constant-bus read; two different pairs in one VOP3P do not assemble).  So no accumulator is
# initialised by a move (16 v_pk_mov per net before), the un-normalisation is packed too, odd layer widths waste nothing, and the
# chains are still acc = fma(W[j][0], x[0], bias[j]); acc = fma(W[j][k], x[k], acc), k ascending: same bits.
# Record layout: np_nets.h::dual_record_len (KBLOB_DUAL, derived from the first layout by np_pack_kblob).
# ------------------------------------------------------------------------------------------------
NM_H = [70, 110, 70]    # activation pairs of layer 1 / 2 / 3 (20, 20, 10 pairs): v70-109, v110-149, and layer 3 over the (dead) layer 1: v70-89
NM_YLO = 90             # v[90:91]: low chain of the output layer, then the net output (A, B) — layer 1's registers are dead by then
NM_YHI = 92             # v[92:93]: high chain
NM_X = 150              # v[150:151], v[152:153], v[154:155]: the (set A, set B) input pairs (live over all records of a class)
NM_ADDR_A, NM_ADDR_B = 156, 157
NM_CLOBBER = list(range(70, 158))


def record_len_nm(IN, H1, H2, H3):
    n = pad2(H1 * (IN + 1)) + pad2(H2 * (H1 + 1))
    n += (pad2(H3 * (H2 + 1)) + pad2(H3 + 1)) if H3 else pad2(H2 + 1)
    n += 2
    return (n + GROUP - 1) // GROUP * GROUP


class BodyNM(Body):
    """One net for both accumulator sets in the neuron-major layout.  Reuses Body's weight-stream bookkeeping (touch / sreg)."""

    def __init__(self, shape, parity, use_next=False):
        super().__init__(shape, parity, use_next)
        self.len = record_len_nm(*shape)
        self.nch = self.len // 16

    def sw(self, pos):
        """(aligned SGPR pair holding record position `pos`, which half)"""
        self.touch(pos)
        r = self.sreg(pos)
        return f's[{r & ~1}:{(r & ~1) + 1}]', r & 1

    @staticmethod
    def pr(r):
        return f'v[{r}:{r + 1}]'

    def dense(self, n_in, n_out, in_base, out_base):
        p0 = self.pos
        assert p0 % 2 == 0
        for k in range(n_in):
            relu = ' clamp' if k == n_in - 1 else ''
            x = self.pr(in_base + 2 * k)
            for j in range(n_out):
                acc = self.pr(out_base + 2 * j)
                if k == 0:      # (W[j][0], bias[j]) in one SGPR pair: acc = fma(w, x, bias) for both sets
                    sp, half = self.sw(p0 + 2 * j)
                    assert half == 0
                    self.ins.append(f'v_pk_fma_f32 {acc}, {sp}, {x}, {sp} op_sel:[0,0,1] op_sel_hi:[0,1,1]{relu}')
                else:
                    sp, half = self.sw(p0 + 2 * n_out + (k - 1) * n_out + j)
                    self.ins.append(f'v_pk_fma_f32 {acc}, {sp}, {x}, {acc} op_sel:[{half},0,0] op_sel_hi:[{half},1,1]{relu}')
        self.pos = p0 + pad2(n_out * (n_in + 1))

    def final(self, n_in, in_base):
        """two interleaved partial chains (numerics spec): lo = bias + even inputs, hi = 0 + odd inputs, y = lo + hi"""
        p0 = self.pos
        assert p0 % 2 == 0 and n_in >= 2
        lo, hi = self.pr(NM_YLO), self.pr(NM_YHI)
        sp, half = self.sw(p0)
        self.ins.append(f'v_pk_fma_f32 {lo}, {sp}, {self.pr(in_base)}, {sp} op_sel:[0,0,1] op_sel_hi:[0,1,1]')
        for k in range(1, n_in):
            sp, half = self.sw(p0 + 1 + k)
            x = self.pr(in_base + 2 * k)
            if k == 1:
                self.ins.append(f'v_pk_fma_f32 {hi}, {sp}, {x}, 0 op_sel:[{half},0,0] op_sel_hi:[{half},1,0]')
            else:
                acc = lo if k % 2 == 0 else hi
                self.ins.append(f'v_pk_fma_f32 {acc}, {sp}, {x}, {acc} op_sel:[{half},0,0] op_sel_hi:[{half},1,1]')
        self.ins.append(f'v_pk_add_f32 {lo}, {lo}, {hi}')
        self.pos = p0 + pad2(n_in + 1)
        # unnormalize: X * std + mean, two roundings (hifi_F16_AeroData.py:36-37), both sets at once
        sp, half = self.sw(self.pos)
        self.ins.append(f'v_pk_mul_f32 {lo}, {sp}, {lo} op_sel:[{half},0] op_sel_hi:[{half},1]')
        sp, half = self.sw(self.pos + 1)
        self.ins.append(f'v_pk_add_f32 {lo}, {sp}, {lo} op_sel:[{half},0] op_sel_hi:[{half},1]')
        self.pos += 2

    def build(self):
        self.dense(self.IN, self.H1, NM_X, NM_H[0])
        self.dense(self.H1, self.H2, NM_H[0], NM_H[1])
        if self.H3:
            self.dense(self.H2, self.H3, NM_H[1], NM_H[2])
            self.final(self.H3, NM_H[2])
        else:
            self.final(self.H2, NM_H[1])
        assert self.pos <= self.len, (self.pos, self.len)
        self.touch(self.len - 1)
        assert self.cur_group == self.len // GROUP - 1
        return self.ins


def gen_function_dual(shape):
    """The class body for TWO aircraft per lane ("pair" kernel variant): one weight stream, two accumulator sets.  Set A = the
    lane's own aircraft, set B = the aircraft of the same lane in the other wave of the workgroup (inputs and output column
    come from / go to LDS); every s_load_dwordx16 now feeds 16 instead of 8 v_pk_fma_f32."""
    IN, H1, H2, H3 = shape
    ln = record_len_nm(*shape)
    ngroups = ln // GROUP
    two_parities = ngroups % 2 == 1
    name = f'mlp_class_asm_dual_{IN}_{H1}_{H2}_{H3}'
    lines = []
    A = lines.append
    A('template <int COUNT, int LDS_STEP>')
    A(f'__device__ __forceinline__ void {name}(const float *w, unsigned lds_addr_a, unsigned lds_addr_b, float x0a, float x1a, float x2a,')
    A('                                                     float x0b, float x1b, float x2b) {')
    A('    asm volatile(')

    def emit(s):
        A(f'        "{s}\\n\\t"')

    emit(f's_mov_b64 s[{S_BASE}:{S_BASE + 1}], %[w]')
    emit(f's_mov_b32 {S_CNT}, %[cnt]')
    for sfx, off in (('a', 0), ('b', 1)):
        for k in range(IN):
            emit(f'v_mov_b32 v{NM_X + 2 * k + off}, %[x{k}{sfx}]')
    emit(f'v_mov_b32 v{NM_ADDR_A}, %[addra]')
    emit(f'v_mov_b32 v{NM_ADDR_B}, %[addrb]')
    for c in range(CPG):
        emit(f's_load_dwordx16 s[{S_W0 + 16 * c}:{S_W0 + 16 * c + 15}], s[{S_BASE}:{S_BASE + 1}], 0x{c * 64:x}')
    emit('.LNP_LOOP_%=:')
    for parity in ([0, 1] if two_parities else [0]):
        for ins in BodyNM(shape, parity).build():
            emit(ins)
        emit(f'ds_write_b32 v{NM_ADDR_A}, v{NM_YLO}')
        emit(f'ds_write_b32 v{NM_ADDR_B}, v{NM_YLO + 1}')
        emit(f'v_add_u32 v{NM_ADDR_A}, %[step], v{NM_ADDR_A}')
        emit(f'v_add_u32 v{NM_ADDR_B}, %[step], v{NM_ADDR_B}')
        emit(f's_add_u32 s{S_BASE}, s{S_BASE}, 0x{ln * 4:x}')
        emit(f's_addc_u32 s{S_BASE + 1}, s{S_BASE + 1}, 0')
        emit(f's_sub_u32 {S_CNT}, {S_CNT}, 1')
        emit(f's_cmp_lg_u32 {S_CNT}, 0')
        if two_parities and parity == 0:
            emit('s_cbranch_scc0 .LNP_DONE_%=')
        else:
            emit('s_cbranch_scc1 .LNP_LOOP_%=')
    emit('.LNP_DONE_%=:')
    emit('s_waitcnt lgkmcnt(0)')
    A('        :')
    A('        : [w] "s"(w), [cnt] "n"(COUNT), [addra] "v"(lds_addr_a), [addrb] "v"(lds_addr_b), [step] "n"(LDS_STEP), [x0a] "v"(x0a), [x1a] "v"(x1a),')
    A('          [x2a] "v"(x2a), [x0b] "v"(x0b), [x1b] "v"(x1b), [x2b] "v"(x2b)')
    regs = NM_CLOBBER
    clob = ', '.join([f'"v{r}"' for r in regs] + [f'"s{r}"' for r in S_CLOBBER] + ['"vcc"', '"scc"', '"memory"'])
    A(f'        : {clob});')
    A('}')
    A('')
    return lines, name


# ------------------------------------------------------------------------------------------------
# Pair variant: phase statements with two accumulator sets.  PAIR_PLANS mirrors np_f16_device.h::PAIR_* (checked by
# static_asserts in the generated file): per phase and per wave of the pair, the (class, first, count) runs it evaluates.
# ------------------------------------------------------------------------------------------------
PAIR_PLANS = {
    'REST': ([('CL_DAMP', 4, 8), ('CL_DLEF', 2, 5), ('CL_E_RUD', 1, 3)],
             [('CL_D_RUD', 1, 1), ('CL_D_LEF', 1, 1), ('CL_E_LEF', 2, 2), ('CL_F', 1, 2), ('CL_C', 0, 5), ('CL_ETA', 0, 1)]),
    'ALL': ([('CL_DAMP', 0, 12), ('CL_DLEF', 0, 7), ('CL_E_LEF', 0, 4), ('CL_YPLEF', 0, 1)],
            [('CL_D_RUD', 0, 2), ('CL_D_LEF', 0, 2), ('CL_E_RUD', 0, 4), ('CL_F', 0, 3), ('CL_YA20', 0, 1), ('CL_C', 0, 5), ('CL_ETA', 0, 1)]),
    'FORCE2': ([('CL_DAMP', 0, 4), ('CL_DLEF', 0, 2), ('CL_E_LEF', 0, 2), ('CL_E_RUD', 0, 1)],
               [('CL_D_RUD', 0, 1), ('CL_D_LEF', 0, 1), ('CL_F', 0, 1), ('CL_YPLEF', 0, 1), ('CL_YA20', 0, 1), ('CL_C', 0, 2)]),
    # aero_1d_tables mode: the 22 single-input nets are table lookups (each wave for its own aircraft, np_f16_device.h::eval_class_pwl);
    # only the multi-input nets go through the two-set bodies.  Balanced by VALU instructions of the two-set bodies (253 / 298 / 653 / 353 / 273 per net).
    'T_REST': ([('CL_D_RUD', 1, 1), ('CL_D_LEF', 1, 1), ('CL_E_LEF', 2, 2), ('CL_F', 1, 1), ('CL_C', 0, 2)],
               [('CL_E_RUD', 1, 3), ('CL_F', 2, 1), ('CL_C', 2, 3)]),
    'T_ALL': ([('CL_D_LEF', 0, 1), ('CL_E_LEF', 0, 4), ('CL_E_RUD', 0, 4), ('CL_YA20', 0, 1), ('CL_C', 0, 2)],
              [('CL_D_RUD', 0, 2), ('CL_D_LEF', 1, 1), ('CL_F', 0, 3), ('CL_C', 2, 3)]),
    'T_FORCE2': ([('CL_D_LEF', 0, 1), ('CL_E_RUD', 0, 1), ('CL_F', 0, 1), ('CL_C', 0, 1)],
                 [('CL_D_RUD', 0, 1), ('CL_E_LEF', 0, 2), ('CL_YA20', 0, 1), ('CL_C', 1, 1)]),
}


def gen_phase_dual(kind, wave):
    CI = {c[0]: i for i, c in enumerate(CLASSES)}
    items = [(CI[c], first, n) for c, first, n in PAIR_PLANS[kind][wave]]
    name = f'mlp_phase_asm_dual_{kind}_{wave}'
    lines = []
    A = lines.append
    A(f'// pair phase {kind}, wave {wave}: ' + ', '.join(f'{CLASSES[ci][0]}[{first}:{first + n}]' for ci, first, n in items))
    A('template <int LDS_STEP>')
    A(f'__device__ __forceinline__ void {name}(const float *w, unsigned lds_base_a, unsigned lds_base_b) {{')
    A('    asm volatile(')

    def emit(s):
        A(f'        "{s}\\n\\t"')

    used_x = sorted({G[g] for ci, _, _ in items for g in CLASSES[ci][2]})
    # the record pointer is the only scalar operand: with s4-s101 and vcc clobbered it (and whatever the compiler keeps in scalar
    # registers across the statement) has s0-s3 to live in — which is why this statement has no second pointer pair
    emit(f's_mov_b64 s[{S_BASE}:{S_BASE + 1}], %[w]')
    for c in range(CPG):
        emit(f's_load_dwordx16 s[{S_W0 + 16 * c}:{S_W0 + 16 * c + 15}], s[{S_BASE}:{S_BASE + 1}], 0x{c * 64:x}')
    parity = 0
    for idx, (ci, first, n) in enumerate(items):
        cname, shape, grps, _, _ = CLASSES[ci]
        ln = record_len_nm(*shape)
        ngroups = ln // GROUP
        start = class_base_nm(ci) + first * ln
        has_next = idx + 1 < len(items)
        if has_next:
            nci, nfirst, _ = items[idx + 1]
            nstart = class_base_nm(nci) + nfirst * record_len_nm(*CLASSES[nci][1])
            delta = nstart - (start + (n - 1) * ln)
            assert delta > 0, (kind, cname, delta)
        emit(f's_mov_b32 {S_CNT}, {n}')
        for k, g in enumerate(grps):
            # the normalised inputs of both sets come straight from their LDS columns (NUM_LIVE + group): no VGPR operands, the
            # first `s_waitcnt lgkmcnt(0)` of the class's first record (records start on group boundaries) covers the reads
            emit(f'ds_read_b32 v{NM_X + 2 * k}, %[addra] offset:%[step]*{NUM_LIVE + G[g]}')
            emit(f'ds_read_b32 v{NM_X + 2 * k + 1}, %[addrb] offset:%[step]*{NUM_LIVE + G[g]}')
        emit(f'v_add_u32 v{NM_ADDR_A}, %[step]*{class_slot(ci) + first}, %[addra]')
        emit(f'v_add_u32 v{NM_ADDR_B}, %[step]*{class_slot(ci) + first}, %[addrb]')
        emit(f'.LNP_L{idx}_%=:')
        pars = [parity, 1 - parity] if (ngroups % 2 == 1 and n > 1) else [parity]
        for pi, par in enumerate(pars):
            emit(f's_mov_b32 vcc_hi, 0x{ln * 4:x}')          # bytes to the record that follows in the sequence
            if has_next:
                emit(f's_cmp_eq_u32 {S_CNT}, 1')
                emit(f's_cmov_b32 vcc_hi, 0x{delta * 4:x}')
            body = BodyNM(shape, par, use_next='advance')
            for ins in body.build():
                emit(ins)
            assert body.advanced
            emit(f'ds_write_b32 v{NM_ADDR_A}, v{NM_YLO}')
            emit(f'ds_write_b32 v{NM_ADDR_B}, v{NM_YLO + 1}')
            emit(f'v_add_u32 v{NM_ADDR_A}, %[step], v{NM_ADDR_A}')
            emit(f'v_add_u32 v{NM_ADDR_B}, %[step], v{NM_ADDR_B}')
            emit(f's_sub_u32 {S_CNT}, {S_CNT}, 1')
            emit(f's_cmp_lg_u32 {S_CNT}, 0')
            if len(pars) == 2 and pi == 0:
                emit(f's_cbranch_scc0 .LNP_D{idx}_%=')
            else:
                emit(f's_cbranch_scc1 .LNP_L{idx}_%=')
        emit(f'.LNP_D{idx}_%=:')
        parity = (parity + n * ngroups) % 2
    emit('s_waitcnt lgkmcnt(0)')
    A('        :')
    ops = '[w] "s"(w), [addra] "v"(lds_base_a), [addrb] "v"(lds_base_b), [step] "n"(LDS_STEP)'
    A(f'        : {ops}')
    regs = NM_CLOBBER
    clob = ', '.join([f'"v{r}"' for r in regs] + [f'"s{r}"' for r in S_CLOBBER] + ['"vcc"', '"scc"', '"memory"'])
    A(f'        : {clob});')
    A('}')
    first_off = class_base_nm(items[0][0]) + items[0][1] * record_len_nm(*CLASSES[items[0][0]][1])
    A(f'constexpr int MLP_PAIR_{kind}_{wave}_START = {first_off};  // KBLOB_DUAL offset of the first record')
    A('')
    return lines


def pair_plan_checks():
    out = ['// the pair plans the dual phase statements were generated from (checked against np_f16_device.h::PAIR_*)']
    for kind, waves in PAIR_PLANS.items():
        for w, items in enumerate(waves):
            conds = [f'PAIR_{kind}.it[{w}][{k}].cl == {c} && PAIR_{kind}.it[{w}][{k}].first == {f} && PAIR_{kind}.it[{w}][{k}].cnt == {n}'
                     for k, (c, f, n) in enumerate(items)]
            conds += [f'PAIR_{kind}.it[{w}][{k}].cnt == 0' for k in range(len(items), 7)]
            out.append(f'#define NPF16_PAIR_PLAN_CHECK_{kind}_{w} ({" && ".join(conds)})')
    return out + ['']


def gen_dual_file():
    out = ['// GENERATED by tools/gen_mlp_asm.py (gen_function_dual) — do not edit.',
           '// Class bodies with two accumulator sets per weight stream (the "pair" kernel variant, np_f16_device.h).', '#pragma once', '']
    for shape in SHAPES:
        lines, _ = gen_function_dual(shape)
        out += lines
    out.append('// record lengths of the two-set layout the generator assumed (checked against np_nets.h::dual_record_len)')
    for IN, H1, H2, H3 in SHAPES:
        out.append(f'static_assert(dual_record_len({IN}, {H1}, {H2}, {H3}) == {record_len_nm(IN, H1, H2, H3)}, "KBLOB_DUAL record layout");')
    out.append('')
    out.append('template <int IN, int H1, int H2, int H3, int COUNT, int LDS_STEP>')
    out.append('__device__ __forceinline__ void mlp_class_asm_dual(const float *w, unsigned addr_a, unsigned addr_b, float x0a, float x1a, float x2a, float x0b,')
    out.append('                                                   float x1b, float x2b) {')
    for i, (IN, H1, H2, H3) in enumerate(SHAPES):
        kw = 'if' if i == 0 else 'else if'
        out.append(f'    {kw} constexpr (IN == {IN} && H1 == {H1} && H2 == {H2} && H3 == {H3}) mlp_class_asm_dual_{IN}_{H1}_{H2}_{H3}<COUNT, LDS_STEP>(w, addr_a, addr_b, x0a, x1a, x2a, x0b, x1b, x2b);')
    out.append('    else static_assert(IN < 0, "no generated dual body for this shape");')
    out.append('}')
    out.append('')
    for kind in PAIR_PLANS:
        for wave in (0, 1):
            out += gen_phase_dual(kind, wave)
    out += pair_plan_checks()
    with open(os.path.join(CSRC, 'np_mlp_asm_dual.inc'), 'w') as f:
        f.write('\n'.join(out) + '\n')
    print('wrote np_mlp_asm_dual.inc')


def gen_function(shape):
    IN, H1, H2, H3 = shape
    ln = record_len(*shape)
    ngroups = ln // GROUP
    two_parities = ngroups % 2 == 1
    name = f'mlp_class_asm_{IN}_{H1}_{H2}_{H3}'
    lines = []
    A = lines.append
    A(f'// shape {IN}-{H1}-{H2}' + (f'-{H3}' if H3 else '') + f'-1: record {ln} floats = {ngroups} groups'
      + (' (odd: two buffer parities)' if two_parities else ''))
    A('template <int COUNT, int LDS_STEP>')
    A(f'__device__ __forceinline__ void {name}(const float *w, unsigned lds_addr, float x0, float x1, float x2) {{')
    A('    asm volatile(')

    def emit(s):
        A(f'        "{s}\\n\\t"')

    emit(f's_mov_b64 s[{S_BASE}:{S_BASE + 1}], %[w]')
    emit(f's_mov_b32 {S_CNT}, %[cnt]')
    emit(f'v_mov_b32 v{V_X}, %[x0]')
    if IN > 1:
        emit(f'v_mov_b32 v{V_X + 2}, %[x1]')
    if IN > 2:
        emit(f'v_mov_b32 v{V_X + 4}, %[x2]')
    emit(f'v_mov_b32 v{V_ADDR}, %[addr]')
    # prologue: first group of the first net into the first CPG buffers
    for c in range(CPG):
        emit(f's_load_dwordx16 s[{S_W0 + 16 * c}:{S_W0 + 16 * c + 15}], s[{S_BASE}:{S_BASE + 1}], 0x{c * 64:x}')
    emit('.LNP_LOOP_%=:')
    for parity in ([0, 1] if two_parities else [0]):
        for ins in Body(shape, parity).build():
            emit(ins)
        emit(f'ds_write_b32 v{V_ADDR}, v{V_Y}')
        emit(f'v_add_u32 v{V_ADDR}, %[step], v{V_ADDR}')
        emit(f's_add_u32 s{S_BASE}, s{S_BASE}, 0x{ln * 4:x}')
        emit(f's_addc_u32 s{S_BASE + 1}, s{S_BASE + 1}, 0')
        emit(f's_sub_u32 {S_CNT}, {S_CNT}, 1')
        emit(f's_cmp_lg_u32 {S_CNT}, 0')
        if two_parities and parity == 0:
            emit('s_cbranch_scc0 .LNP_DONE_%=')
        else:
            emit('s_cbranch_scc1 .LNP_LOOP_%=')
    emit('.LNP_DONE_%=:')
    emit('s_waitcnt lgkmcnt(0)')  # retire the dangling prefetch of the record after the last one
    A('        :')
    A('        : [w] "s"(w), [cnt] "n"(COUNT), [addr] "v"(lds_addr), [step] "n"(LDS_STEP), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2)')
    clob = ', '.join([f'"v{r}"' for r in V_CLOBBER] + [f'"s{r}"' for r in S_CLOBBER] + ['"vcc"', '"scc"', '"memory"'])
    A(f'        : {clob});')
    A('}')
    A('')
    return lines, name



# ------------------------------------------------------------------------------------------------
# Phase functions: ONE asm statement for a whole sequence of classes (np_nets.h::CLASSES order).
# Inside a class the records are contiguous; between classes the stream jumps.  Every net body
# prefetches "the first group of the record that follows in the sequence" through S_NEXT, which is
# the next record of the class or, for the last net of a class, the first record of the next class
# (a compile-time byte delta).  So the weight stream never drains between classes: one prologue and
# one epilogue per PHASE instead of per class (19 -> 3 exposed scalar-load latencies per env.step).
# ------------------------------------------------------------------------------------------------
G = {'G_A_C': 0, 'G_A_DAMP': 1, 'G_A_LEF': 2, 'G_A_DLEF': 3, 'G_A_RUD': 4, 'G_B_C': 5, 'G_B_O': 6, 'G_E_C': 7, 'G_E_ETA': 8}
KBLOB_HEADER = 3 * len(G) + 1   # np_nets.h: (mean, sigma, 1/sigma) per normalisation group, padded to even
X_IN_LDS = False  # True: the phase statements read their inputs from LDS slots NUM_LIVE.. (51 slots/lane: only 5 workgroups per CU)
NUM_LIVE = 42    # output slots; the normalised inputs live in LDS slots NUM_LIVE .. NUM_LIVE + 8 (np_nets.h::NUM_LDS_SLOTS)
# (name, shape, input groups, count, n_force) — mirror of np_nets.h::CLASSES, cross-checked by static_asserts in the output
CLASSES = [
    ('CL_DAMP', (1, 20, 10, 0), ('G_A_DAMP',), 12, 4),
    ('CL_DLEF', (1, 20, 10, 0), ('G_A_DLEF',), 7, 2),
    ('CL_D_RUD', (2, 20, 10, 0), ('G_A_RUD', 'G_B_O'), 2, 1),
    ('CL_D_LEF', (2, 20, 10, 0), ('G_A_LEF', 'G_B_O'), 2, 1),
    ('CL_E_LEF', (2, 20, 10, 5), ('G_A_LEF', 'G_B_O'), 4, 2),
    ('CL_E_RUD', (2, 20, 10, 5), ('G_A_RUD', 'G_B_O'), 4, 1),
    ('CL_F', (2, 20, 20, 10), ('G_A_LEF', 'G_B_O'), 3, 1),
    ('CL_YPLEF', (1, 20, 10, 5), ('G_A_DLEF',), 1, 1),
    ('CL_YA20', (2, 20, 10, 10), ('G_A_RUD', 'G_B_O'), 1, 1),
    ('CL_C', (3, 20, 10, 0), ('G_A_C', 'G_B_C', 'G_E_C'), 5, 2),
    ('CL_ETA', (1, 20, 10, 0), ('G_E_ETA',), 1, 0),
]
NUM_AB = 9


def class_base(ci):
    return KBLOB_HEADER + sum(c[3] * record_len(*c[1]) for c in CLASSES[:ci])


def class_base_nm(ci):
    return sum(c[3] * record_len_nm(*c[1]) for c in CLASSES[:ci])


def class_slot(ci):
    return sum(c[3] for c in CLASSES[:ci])


def phase_items(kind):
    if kind.startswith('T_'):   # aero_1d_tables mode: the multi-input nets of the phase only (the others are table lookups)
        return [(ci, first, n) for ci, first, n in phase_items(kind[2:]) if CLASSES[ci][1][0] > 1 and n > 0]
    ab = range(NUM_AB)
    if kind == 'ALL':
        items = [(ci, 0, CLASSES[ci][3]) for ci in ab] + [(9, 0, 5), (10, 0, 1)]
    elif kind == 'REST':
        items = [(ci, CLASSES[ci][4], CLASSES[ci][3] - CLASSES[ci][4]) for ci in ab] + [(9, 0, 5), (10, 0, 1)]
    elif kind == 'FORCE2':
        items = [(ci, 0, CLASSES[ci][4]) for ci in ab] + [(9, 0, 2)]
    else:
        raise ValueError(kind)
    return [it for it in items if it[2] > 0]


def gen_phase(kind):
    items = phase_items(kind)
    name = f'mlp_phase_asm_{kind}'
    lines = []
    A = lines.append
    A(f'// phase {kind}: ' + ', '.join(f'{CLASSES[ci][0]}[{first}:{first + n}]' for ci, first, n in items))
    A('template <int LDS_STEP>')
    if X_IN_LDS:
        A(f'__device__ __forceinline__ void {name}(const float *w, unsigned lds_base) {{')
    else:
        A(f'__device__ __forceinline__ void {name}(const float *w, unsigned lds_base, const float (&xn)[{len(G)}]) {{')
    A('    asm volatile(')

    def emit(s):
        A(f'        "{s}\\n\\t"')

    used_x = sorted({G[g] for ci, _, _ in items for g in CLASSES[ci][2]})
    emit(f's_mov_b64 s[{S_BASE}:{S_BASE + 1}], %[w]')
    shape0 = CLASSES[items[0][0]][1]
    for c in range(CPG):   # prologue: first group of the first net of the first class
        emit(f's_load_dwordx16 s[{S_W0 + 16 * c}:{S_W0 + 16 * c + 15}], s[{S_BASE}:{S_BASE + 1}], 0x{c * 64:x}')
    parity = 0
    for idx, (ci, first, n) in enumerate(items):
        cname, shape, grps, _, _ = CLASSES[ci]
        ln = record_len(*shape)
        ngroups = ln // GROUP
        start = class_base(ci) + first * ln
        has_next = idx + 1 < len(items)
        if has_next:
            nci, nfirst, _ = items[idx + 1]
            nstart = class_base(nci) + nfirst * record_len(*CLASSES[nci][1])
            delta = nstart - (start + (n - 1) * ln)
            assert delta > 0, (kind, cname, delta)
        emit(f's_mov_b32 {S_CNT}, {n}')
        for k, g in enumerate(grps):
            if X_IN_LDS:   # from this lane's LDS slots NUM_LIVE + g (retired by the first group wait)
                emit(f'ds_read_b32 v{V_X + 2 * k}, %[addr] offset:%[step]*{NUM_LIVE + G[g]}')
            else:
                emit(f'v_mov_b32 v{V_X + 2 * k}, %[x{G[g]}]')
        emit(f'v_add_u32 v{V_ADDR}, %[step]*{class_slot(ci) + first}, %[addr]')
        emit(f'.LNP_L{idx}_%=:')
        pars = [parity, 1 - parity] if (ngroups % 2 == 1 and n > 1) else [parity]
        for pi, par in enumerate(pars):
            emit(f's_mov_b32 s{S_NEXT}, 0x{ln * 4:x}')
            if has_next:
                emit(f's_cmp_eq_u32 {S_CNT}, 1')
                emit(f's_cmov_b32 s{S_NEXT}, 0x{delta * 4:x}')
            emit(f's_add_u32 s{S_NEXT}, s{S_BASE}, s{S_NEXT}')
            emit(f's_addc_u32 s{S_NEXT + 1}, s{S_BASE + 1}, 0')
            for ins in Body(shape, par, use_next=True).build():
                emit(ins)
            emit(f'ds_write_b32 v{V_ADDR}, v{V_Y}')
            emit(f'v_add_u32 v{V_ADDR}, %[step], v{V_ADDR}')
            emit(f's_mov_b64 s[{S_BASE}:{S_BASE + 1}], s[{S_NEXT}:{S_NEXT + 1}]')
            emit(f's_sub_u32 {S_CNT}, {S_CNT}, 1')
            emit(f's_cmp_lg_u32 {S_CNT}, 0')
            if len(pars) == 2 and pi == 0:
                emit(f's_cbranch_scc0 .LNP_D{idx}_%=')
            else:
                emit(f's_cbranch_scc1 .LNP_L{idx}_%=')
        emit(f'.LNP_D{idx}_%=:')
        parity = (parity + n * ngroups) % 2
    emit('s_waitcnt lgkmcnt(0)')  # retire the dangling prefetch after the last record of the phase
    A('        :')
    ops = '[w] "s"(w), [addr] "v"(lds_base), [step] "n"(LDS_STEP)'
    if not X_IN_LDS:
        ops += ', ' + ', '.join(f'[x{k}] "v"(xn[{k}])' for k in used_x)
    A(f'        : {ops}')
    clob = ', '.join([f'"v{r}"' for r in V_CLOBBER] + [f'"s{r}"' for r in S_CLOBBER + [S_NEXT, S_NEXT + 1]] + ['"vcc"', '"scc"', '"memory"'])
    A(f'        : {clob});')
    A('}')
    A('')
    first_off = class_base(items[0][0]) + items[0][1] * record_len(*CLASSES[items[0][0]][1])
    A(f'constexpr int MLP_PHASE_{kind}_START = {first_off};  // KBLOB offset of the first record of the phase')
    A('')
    return lines


def phase_checks():
    out = ['// the class table the phase functions were generated from (checked against np_nets.h::CLASSES)']
    for ci, (cname, shape, grps, count, nforce) in enumerate(CLASSES):
        IN, H1, H2, H3 = shape
        conds = [f'CLASSES[{cname}].n_in == {IN}', f'CLASSES[{cname}].h1 == {H1}', f'CLASSES[{cname}].h2 == {H2}', f'CLASSES[{cname}].h3 == {H3}',
                 f'CLASSES[{cname}].count == {count}', f'CLASSES[{cname}].n_force == {nforce}', f'class_base({cname}) == {class_base(ci)}',
                 f'class_slot({cname}) == {class_slot(ci)}', f'(int){cname} == {ci}']
        conds += [f'CLASSES[{cname}].grp[{k}] == {g}' for k, g in enumerate(grps)]
        out.append(f'static_assert({" && ".join(conds)}, "phase asm: class table mismatch ({cname})");')
    out.append(f'static_assert(KBLOB_HEADER == {KBLOB_HEADER} && NUM_AB_CLASSES == {NUM_AB} && NUM_LIVE_NETS == {NUM_LIVE} && '
               f'NUM_LDS_SLOTS == {NUM_LIVE + (len(G) if X_IN_LDS else 0)}, "phase asm: KBLOB header / class split / LDS slots");')
    out.append(f'#define NPF16_PHASE_X_IN_LDS {1 if X_IN_LDS else 0}')
    return out + ['']


# ------------------------------------------------------------------------------------------------
# The fused low-level controller (np_actor.h): 16 outputs of a Linear(128, .) layer for one wave.  Same weight-stream scheme:
# the packed weights are k-major, so the 16 weights of input feature k are ONE s_load_dwordx16; groups of 3 input features are
# double-buffered in s[4:51] / s[52:99]; the per-lane inputs x[k] come from the LDS matrix [feature][lane] by ds_read_b32 issued
# together with the loads of their group (one s_waitcnt lgkmcnt(0) retires both).  acc = bias; acc = fma(W[j][k], x[k], acc),
# k ascending — bit-identical to the C++ loop it replaces.
# ------------------------------------------------------------------------------------------------
ACTOR_OUT = os.path.join(CSRC, 'np_actor_asm.inc')
A_ACC = 70        # v[70:85] accumulators
A_X = [86, 87, 88, 89, 90, 91]   # inputs of the two groups in flight
A_ADDR = 92       # LDS byte address of x[k = first feature of the iteration] for this lane
A_ROW = 256       # bytes between two features in the LDS matrix (64 lanes x 4)


def gen_actor_dense(K=128):
    lines = []
    A = lines.append
    A('// GENERATED by tools/gen_mlp_asm.py (gen_actor_dense) — do not edit.')
    A('#pragma once')
    A(f'// 16 outputs of a Linear({K}, .) layer: w = &Wt[0][j0] (rows LD_BYTES apart), b = &bias[j0], xaddr = LDS byte address of x[0] for this lane')
    A('template <int LD_BYTES>')
    A('__device__ __forceinline__ void actor_dense16_asm(const float *w, const float *b, unsigned xaddr, float (&acc)[16]) {')
    A('    asm volatile(')

    def emit(s):
        A(f'        "{s}\\n\\t"')

    def sload(buf_set, slot, chunk):   # chunk = feature index relative to the moving base
        r = S_W0 + 48 * buf_set + 16 * slot
        emit(f's_load_dwordx16 s[{r}:{r + 15}], s[{S_BASE}:{S_BASE + 1}], %[ld]*{chunk}')

    def xread(reg, feat):
        emit(f'ds_read_b32 v{reg}, v{A_ADDR} offset:{A_ROW * feat}')

    def compute(buf_set, slot, xreg):
        r = S_W0 + 48 * buf_set + 16 * slot
        e = xreg & 1
        xp = f'v[{xreg - e}:{xreg - e + 1}]'
        for p in range(8):
            acc = f'v[{A_ACC + 2 * p}:{A_ACC + 2 * p + 1}]'
            emit(f'v_pk_fma_f32 {acc}, s[{r + 2 * p}:{r + 2 * p + 1}], {xp}, {acc} op_sel:[0,{e},0] op_sel_hi:[1,{e},1]')

    iters, tail = divmod(K, 6)
    assert tail in (0, 2)
    emit(f's_mov_b64 s[{S_BASE}:{S_BASE + 1}], %[w]')
    emit(f'v_mov_b32 v{A_ADDR}, %[xaddr]')
    emit(f's_load_dwordx16 s[{S_W0 + 48}:{S_W0 + 63}], %[b], 0x0')        # bias -> first chunk of set 1
    for s in range(3):
        sload(0, s, s)
        xread(A_X[s], s)
    emit('s_waitcnt lgkmcnt(0)')
    for p in range(8):
        sp = f's[{S_W0 + 48 + 2 * p}:{S_W0 + 48 + 2 * p + 1}]'
        emit(f'v_pk_mov_b32 v[{A_ACC + 2 * p}:{A_ACC + 2 * p + 1}], {sp}, {sp} op_sel:[0,1]')
    emit(f's_mov_b32 {S_CNT}, {iters}')
    emit('.LNA_LOOP_%=:')
    for s in range(3):                 # group B of this iteration: features 3..5 -> set 1
        sload(1, s, 3 + s)
        xread(A_X[3 + s], 3 + s)
    for s in range(3):
        compute(0, s, A_X[s])
    emit('s_waitcnt lgkmcnt(0)')
    for s in range(3):                 # group A of the next iteration: features 6..8 -> set 0 (the last one runs past K: harmless)
        sload(0, s, 6 + s)
        xread(A_X[s], 6 + s)
    for s in range(3):
        compute(1, s, A_X[3 + s])
    emit(f's_add_u32 s{S_BASE}, s{S_BASE}, %[ld]*6')
    emit(f's_addc_u32 s{S_BASE + 1}, s{S_BASE + 1}, 0')
    emit(f'v_add_u32 v{A_ADDR}, {A_ROW * 6}, v{A_ADDR}')
    emit('s_waitcnt lgkmcnt(0)')
    emit(f's_sub_u32 {S_CNT}, {S_CNT}, 1')
    emit(f's_cmp_lg_u32 {S_CNT}, 0')
    emit('s_cbranch_scc1 .LNA_LOOP_%=')
    for s in range(tail):              # the last K % 6 features sit in set 0 already
        compute(0, s, A_X[s])
    for j in range(16):
        emit(f'v_mov_b32 %[o{j}], v{A_ACC + j}')
    outs = ', '.join(f'[o{j}] "=v"(acc[{j}])' for j in range(16))
    A(f'        : {outs}')
    A('        : [w] "s"(w), [b] "s"(b), [xaddr] "v"(xaddr), [ld] "n"(LD_BYTES)')
    clob = ', '.join([f'"v{r}"' for r in range(A_ACC, A_ADDR + 1)] + [f'"s{r}"' for r in S_CLOBBER] + ['"vcc"', '"scc"', '"memory"'])
    A(f'        : {clob});')
    A('}')
    with open(ACTOR_OUT, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print('wrote', ACTOR_OUT)


# ------------------------------------------------------------------------------------------------
# The fused controller on the matrix cores (np_actor.h, actor_forward_mfma_kernel): one wave, 32 output features x 64 rows of a
# Linear(128, .) layer as a chain of 128 v_mfma_f32_32x32x1_2b_f32 (K = 1 each: one exact fused multiply-add per element, so the
# chain is acc = fma(W[f][k], x[k], acc), k ascending).  Operands are fetched a group of M_G features ahead: A = weights by
# global_load_dword (SGPR base + per-lane offset + immediate), B = inputs by ds_read2st64_b32 from the LDS matrix [feature][row].
# ------------------------------------------------------------------------------------------------
M_G = 8
M_A = [222, 238]          # v[222:229] / v[238:245]: A operands of the two groups in flight
M_B = [230, 246]          # v[230:237] / v[246:253]: B operands
M_VOFF = 254              # per-lane byte offset of the weight column
M_XADDR = 255             # LDS byte address of x[first feature of the group][lane]
M_SBASE = 4               # s[4:5], s[6:7], s[8:9]: row bases (a 12-bit immediate reaches 4095 B)
M_SCNT = 10
M_SNEXT = 12              # s[12:17]: row bases of the next layer's first group


def gen_actor_mfma(lines, LD, LDN, K=128):
    A = lines.append
    stride = LD * 4
    per_base = min(M_G, 4095 // stride + 1)
    nbase = (M_G + per_base - 1) // per_base
    assert nbase <= 3 and K % (2 * M_G) == 0
    A(f'// 32 features x 64 rows of a Linear({K}, .) layer whose packed rows are {LD} floats apart: w = &Wt[0][f0] (wave-uniform),')
    A('// voff = 4 * (lane % 32), xaddr = LDS byte address of x[0][lane]; pa = W[f][0..7] of this lane (prefetched); acc holds the bias on entry')
    A(f'// wnext = &Wt[0][f0] of the layer that runs next (rows {LDN} floats apart): its first 8 A operands are fetched behind the last')
    A('// MFMAs of this layer and returned in pa, so the next chain starts without waiting for L2')
    A('template <>')
    A(f'__device__ __forceinline__ void actor_dense_mfma_asm<{LD}, {LDN}>(const float *w, const float *wnext, unsigned voff, unsigned xaddr,')
    A('                                                                  float (&pa)[8], f32x32 &acc) {')
    A('    asm volatile(')

    def emit(s):
        A(f'        "{s}\\n\\t"')

    def loads(g):      # the next M_G features into register group g, then advance the bases and the LDS address
        for u in range(M_G):
            b, r = divmod(u, per_base)
            sb = M_SBASE + 2 * b
            emit(f'global_load_dword v{M_A[g] + u}, v{M_VOFF}, s[{sb}:{sb + 1}] offset:{r * stride}')
        for u in range(0, M_G, 2):
            emit(f'ds_read2st64_b32 v[{M_B[g] + u}:{M_B[g] + u + 1}], v{M_XADDR} offset0:{u} offset1:{u + 1}')
        for b in range(nbase):
            sb = M_SBASE + 2 * b
            emit(f's_add_u32 s{sb}, s{sb}, {M_G * stride}')
            emit(f's_addc_u32 s{sb + 1}, s{sb + 1}, 0')
        emit(f'v_add_u32 v{M_XADDR}, {M_G * 256}, v{M_XADDR}')

    def mfmas(g, lo, hi):
        for u in range(lo, hi):
            emit(f'v_mfma_f32_32x32x1_2b_f32 %[acc], v{M_A[g] + u}, v{M_B[g] + u}, %[acc]')

    def bump_bases():
        for b in range(nbase):
            sb = M_SBASE + 2 * b
            emit(f's_add_u32 s{sb}, s{sb}, {M_G * stride}')
            emit(f's_addc_u32 s{sb + 1}, s{sb + 1}, 0')

    # prologue: the A operands of features 0..7 were fetched by the caller (before the previous layer's epilogue) and arrive as
    # inputs; only the B operands (LDS, short latency) are read here
    emit(f's_mov_b64 s[{M_SBASE}:{M_SBASE + 1}], %[w]')
    for b in range(1, nbase):
        sb = M_SBASE + 2 * b
        emit(f's_add_u32 s{sb}, s{M_SBASE}, {b * per_base * stride}')
        emit(f's_addc_u32 s{sb + 1}, s{M_SBASE + 1}, 0')
    emit(f'v_mov_b32 v{M_VOFF}, %[voff]')
    emit(f'v_mov_b32 v{M_XADDR}, %[xaddr]')
    for u in range(M_G):
        emit(f'v_mov_b32 v{M_A[0] + u}, %[p{u}]')
    for u in range(0, M_G, 2):
        emit(f'ds_read2st64_b32 v[{M_B[0] + u}:{M_B[0] + u + 1}], v{M_XADDR} offset0:{u} offset1:{u + 1}')
    bump_bases()
    emit(f'v_add_u32 v{M_XADDR}, {M_G * 256}, v{M_XADDR}')
    emit(f's_mov_b32 s{M_SCNT}, {K // (2 * M_G) - 1}')
    emit('s_waitcnt lgkmcnt(0)')
    # steady state: the next group's loads are issued behind the first MFMA of the current group (the matrix pipe never drains:
    # the previous user of the registers they overwrite completed before that MFMA could start, the chain is dependent)
    emit('.LMF_LOOP_%=:')
    mfmas(0, 0, 1)
    loads(1)
    mfmas(0, 1, M_G)
    emit('s_waitcnt vmcnt(0) lgkmcnt(0)')
    mfmas(1, 0, 1)
    loads(0)
    mfmas(1, 1, M_G)
    emit(f's_sub_u32 s{M_SCNT}, s{M_SCNT}, 1')
    emit('s_waitcnt vmcnt(0) lgkmcnt(0)')
    emit(f's_cmp_lg_u32 s{M_SCNT}, 0')
    emit('s_cbranch_scc1 .LMF_LOOP_%=')
    mfmas(0, 0, 1)
    loads(1)
    mfmas(0, 1, M_G)
    emit('s_waitcnt vmcnt(0) lgkmcnt(0)')
    mfmas(1, 0, 1)
    nstride = LDN * 4
    nper = min(M_G, 4095 // nstride + 1)
    for b in range((M_G + nper - 1) // nper):
        sb = M_SNEXT + 2 * b
        if b == 0:
            emit(f's_mov_b64 s[{sb}:{sb + 1}], %[wn]')
        else:
            emit(f's_add_u32 s{sb}, s{M_SNEXT}, {b * nper * nstride}')
            emit(f's_addc_u32 s{sb + 1}, s{M_SNEXT + 1}, 0')
    for u in range(M_G):
        b, r = divmod(u, nper)
        sb = M_SNEXT + 2 * b
        emit(f'global_load_dword %[p{u}], v{M_VOFF}, s[{sb}:{sb + 1}] offset:{r * nstride}')
    mfmas(1, 1, M_G)
    emit('s_waitcnt vmcnt(0)')
    emit('s_nop 15')          # 16-pass XDL result -> the VALU / LDS instructions the compiler places after this statement
    emit('s_nop 7')
    outs = ', '.join(f'[p{u}] "+v"(pa[{u}])' for u in range(M_G))   # in: consumed by the v_movs of the prologue; out: the next layer's
    A(f'        : [acc] "+v"(acc), {outs}')
    A('        : [w] "s"(w), [wn] "s"(wnext), [voff] "v"(voff), [xaddr] "v"(xaddr)')
    regs = list(range(M_A[0], M_XADDR + 1))
    clob = ', '.join([f'"v{r}"' for r in regs] + [f'"s{r}"' for r in range(M_SBASE, M_SNEXT + 6)] + ['"scc"', '"memory"'])
    A(f'        : {clob});')
    A('}')


def gen_actor_mfma_file():
    out = os.path.join(CSRC, 'np_actor_mfma_asm.inc')
    lines = ['// GENERATED by tools/gen_mlp_asm.py (gen_actor_mfma) — do not edit.', '#pragma once',
             'typedef float f32x32 __attribute__((ext_vector_type(32)));',
             'template <int LD, int LD_NEXT>',
             '__device__ __forceinline__ void actor_dense_mfma_asm(const float *w, const float *wnext, unsigned voff, unsigned xaddr, float (&pa)[8], f32x32 &acc);']
    for LD in (128, 384):
        for LDN in (128, 384):
            gen_actor_mfma(lines, LD, LDN)
    with open(out, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print('wrote', out)


# ------------------------------------------------------------------------------------------------
# The same layer for the 32-row tile of actor_forward_mfma32_kernel (small batches: twice as many workgroups, half the chain
# time per tile): one wave, 32 features x 32 rows as a chain of v_mfma_f32_16x16x1_4b_f32 (four 16 x 16 blocks: A = features
# f0 + 16 (l / 32) + l % 16, B = rows l % 32).  The bias enters as one more K = 1 step (A = bias, B = 1.0, C = 0), so the chain
# starts without a separate accumulator fill; operands are fetched a group of 16 features ahead (16 x 40 cycles of dependent
# MFMAs cover the L2 latency of the weights with margin).
# ------------------------------------------------------------------------------------------------
N_G = 16                  # features per operand group
N_RING = 3                # A operand groups in flight: group g + 2 is requested behind the first MFMA of group g
N_A = [112, 128, 144]     # v[112:127] / v[128:143] / v[144:159]
N_B = [160, 176]          # v[160:175] / v[176:191]: B operands (LDS: one group ahead is enough)
N_VOFF = 192              # per-lane byte offset of the weight column
N_XADDR = 193             # LDS byte address of x[first feature of the group][row]; v194 = the same + 8 features
N_ONE = 195
N_SBASE = 4               # s[4:5]: running row base of this layer, s[6:7]: of the next layer's prefetch
N_ROWS = 32               # rows per tile = floats per LDS matrix row
N_PRE = 2 * N_G + 1       # values handed from layer to layer: the first two A groups and the bias


def gen_actor_mfma16(lines, LD, LDN, K=128):
    A = lines.append
    stride, nstride = LD * 4, LDN * 4
    NGRP = K // N_G
    assert K % N_G == 0 and NGRP >= 3
    A(f'// 32 features x 32 rows of a Linear({K}, .) layer whose packed rows are {LD} floats apart: w = &Wt[0][f0] (wave-uniform),')
    A('// voff = 4 * (16 * (lane / 32) + lane % 16), xaddr = LDS byte address of x[0][lane % 32]; pa[0..31] = W[f][0..31] of this lane and')
    A('// pa[32] = bias[f] (prefetched); acc is written (not read).  wnext / bnext = &Wt[0][f0] / &bias[f0] of the layer that runs next')
    A(f'// (rows {LDN} floats apart): its first 32 A operands and its bias are fetched behind the MFMAs of the last two groups and returned in pa')
    A('template <>')
    A(f'__device__ __forceinline__ void actor_dense_mfma16_asm<{LD}, {LDN}>(const float *w, const float *wnext, const float *bnext, unsigned voff,')
    A(f'                                                                    unsigned xaddr, float (&pa)[{N_PRE}], f32x16 &acc) {{')
    A('    asm volatile(')

    def emit(s):
        A(f'        "{s}\\n\\t"')

    state = {'off': 0, 'noff': 0}
    queue = []        # tags of the vector loads in issue order (they return in that order): ('A', group) or ('N',)

    def a_loads(g, out):    # A operands of group g (features 16 g ..) into ring slot g % N_RING; the running base absorbs what the immediate cannot reach
        for u in range(N_G):
            if state['off'] > 4095:
                out.append(f's_add_u32 s{N_SBASE}, s{N_SBASE}, {state["off"]}')
                out.append(f's_addc_u32 s{N_SBASE + 1}, s{N_SBASE + 1}, 0')
                state['off'] = 0
            out.append(f'global_load_dword v{N_A[g % N_RING] + u}, v{N_VOFF}, s[{N_SBASE}:{N_SBASE + 1}] offset:{state["off"]}')
            queue.append(('A', g))
            state['off'] += stride

    def next_loads(lo, hi, out):   # the next layer's A operands lo..hi-1 into the pa registers
        for u in range(lo, hi):
            if state['noff'] > 4095:
                out.append(f's_add_u32 s{N_SBASE + 2}, s{N_SBASE + 2}, {state["noff"]}')
                out.append(f's_addc_u32 s{N_SBASE + 3}, s{N_SBASE + 3}, 0')
                state['noff'] = 0
            out.append(f'global_load_dword %[p{u}], v{N_VOFF}, s[{N_SBASE + 2}:{N_SBASE + 3}] offset:{state["noff"]}')
            queue.append(('N',))
            state['noff'] += nstride

    def b_reads(g, out):    # B operands of group g into slot g % 2
        base = N_B[g % 2]
        for u in range(0, N_G, 2):
            addr = N_XADDR + (1 if u >= 8 else 0)
            uu = u % 8
            out.append(f'ds_read2_b32 v[{base + u}:{base + u + 1}], v{addr} offset0:{uu * N_ROWS} offset1:{(uu + 1) * N_ROWS}')
        out.append(f'v_add_u32 v{N_XADDR}, {N_G * N_ROWS * 4}, v{N_XADDR}')
        out.append(f'v_add_u32 v{N_XADDR + 1}, {N_G * N_ROWS * 4}, v{N_XADDR + 1}')

    # prologue: groups 0 and 1 of the A operands and the bias arrive in pa (fetched during the previous layer)
    emit(f's_mov_b64 s[{N_SBASE}:{N_SBASE + 1}], %[w]')
    emit(f's_mov_b64 s[{N_SBASE + 2}:{N_SBASE + 3}], %[wn]')
    emit(f'v_mov_b32 v{N_VOFF}, %[voff]')
    emit(f'v_mov_b32 v{N_XADDR}, %[xaddr]')
    emit(f'v_add_u32 v{N_XADDR + 1}, {8 * N_ROWS * 4}, v{N_XADDR}')
    pro = []
    b_reads(0, pro)
    state['off'] = 2 * N_G * stride
    a_loads(2, pro)
    for x in pro:
        emit(x)
    emit(f'v_mov_b32 v{N_ONE}, 1.0')
    for u in range(N_G):
        emit(f'v_mov_b32 v{N_A[0] + u}, %[p{u}]')
    emit(f'v_mfma_f32_16x16x1_4b_f32 %[acc], %[p{2 * N_G}], v{N_ONE}, 0')      # acc = fma(bias, 1, 0)
    for u in range(N_G):
        emit(f'v_mov_b32 v{N_A[1] + u}, %[p{N_G + u}]')
    emit('s_waitcnt lgkmcnt(0)')
    # other instructions are spread behind the MFMAs, 3 to 5 per gap (the chain is dependent: 32 cycles between two of them)
    for g in range(NGRP):
        # on entry: the A and B operands of group g have landed.  Behind its MFMAs: the B operands of group g + 1 (LDS), the A
        # operands of group g + 2 (into the slot group g - 1 has just left), and in groups NGRP - 3 / NGRP - 2 the next layer's first
        # two groups + bias, so that the statement's final wait finds them landed
        side = []
        if g + 1 < NGRP:
            b_reads(g + 1, side)
        if g >= 1 and g + 2 < NGRP:
            a_loads(g + 2, side)
        if g == NGRP - 3:
            next_loads(0, N_G, side)
            side.append(f'global_load_dword %[p{2 * N_G}], v{N_VOFF}, %[bn]')
            queue.append(('N',))
        if g == NGRP - 2:
            next_loads(N_G, 2 * N_G, side)
        if os.environ.get('NPF16_GEN_ACTOR_NOSIDE') == '1':   # TIMING EXPERIMENT ONLY (wrong results): no operand traffic behind the MFMAs of groups 1.. —
            side = [x for x in side if not (x.startswith('global_load_dword v') or x.startswith('ds_read2') or x.startswith('v_add_u32') or x.startswith('s_add'))]   # what the chain costs without side instructions
            queue[:] = [t for t in queue if t == ('N',)]
        PER_GAP = max(3, -(-len(side) // (N_G - 1)))
        assert PER_GAP <= 5, len(side)
        for u in range(N_G):
            emit(f'v_mfma_f32_16x16x1_4b_f32 %[acc], v{N_A[g % N_RING] + u}, v{N_B[g % 2] + u}, %[acc]')
            for x in side[PER_GAP * u:PER_GAP * (u + 1)]:
                emit(x)
        if g + 1 < NGRP:
            if g + 1 <= 1:
                emit('s_waitcnt lgkmcnt(0)')      # group 1 came in registers
            else:
                # loads still allowed in flight when group g + 1 starts: everything issued after the last of its own A operands
                if os.environ.get('NPF16_GEN_ACTOR_NOSIDE') == '1':
                    emit('s_waitcnt lgkmcnt(0)')
                    continue
                last = max(i for i, t in enumerate(queue) if t == ('A', g + 1))
                emit(f's_waitcnt vmcnt({len(queue) - 1 - last}) lgkmcnt(0)')
    emit('s_waitcnt vmcnt(0)')
    emit('s_nop 15')          # 8-pass XDL result -> the VALU / LDS instructions the compiler places after this statement
    outs = ', '.join(f'[p{u}] "+v"(pa[{u}])' for u in range(N_PRE))
    A(f'        : [acc] "=&v"(acc), {outs}')
    A('        : [w] "s"(w), [wn] "s"(wnext), [bn] "s"(bnext), [voff] "v"(voff), [xaddr] "v"(xaddr)')
    regs = list(range(N_A[0], N_ONE + 1))
    clob = ', '.join([f'"v{r}"' for r in regs] + [f'"s{r}"' for r in range(N_SBASE, N_SBASE + 4)] + ['"scc"', '"memory"'])
    A(f'        : {clob});')
    A('}')


def gen_actor_mfma16_file():
    out = os.path.join(CSRC, 'np_actor_mfma16_asm.inc')
    lines = ['// GENERATED by tools/gen_mlp_asm.py (gen_actor_mfma16) — do not edit.', '#pragma once',
             'typedef float f32x16 __attribute__((ext_vector_type(16)));',
             'template <int LD, int LD_NEXT>',
             '__device__ __forceinline__ void actor_dense_mfma16_asm(const float *w, const float *wnext, const float *bnext, unsigned voff, unsigned xaddr, float (&pa)[33], f32x16 &acc);']
    for LD in (128, 384):
        for LDN in (128, 384):
            gen_actor_mfma16(lines, LD, LDN)
    with open(out, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print('wrote', out)


def main():
    out = ['// GENERATED by tools/gen_mlp_asm.py — do not edit.  See that file for the design notes.',
           '// One asm statement per net class: double-buffered scalar weight stream + v_pk_fma_f32 chains.',
           '#pragma once', '']
    for shape in SHAPES:
        lines, _ = gen_function(shape)
        out += lines
    out.append('// record lengths the generator assumed (checked against np_nets.h::asm_record_len)')
    for shape in SHAPES:
        IN, H1, H2, H3 = shape
        out.append(f'static_assert(asm_record_len({IN}, {H1}, {H2}, {H3}) == {record_len(*shape)}, "KBLOB record layout");')
    out.append('')
    out.append('template <int IN, int H1, int H2, int H3, int COUNT, int LDS_STEP>')
    out.append('__device__ __forceinline__ void mlp_class_asm(const float *w, unsigned lds_addr, float x0, float x1, float x2) {')
    first = True
    for shape in SHAPES:
        IN, H1, H2, H3 = shape
        kw = 'if' if first else 'else if'
        out.append(f'    {kw} constexpr (IN == {IN} && H1 == {H1} && H2 == {H2} && H3 == {H3}) '
                   f'mlp_class_asm_{IN}_{H1}_{H2}_{H3}<COUNT, LDS_STEP>(w, lds_addr, x0, x1, x2);')
        first = False
    out.append('    else static_assert(IN < 0, "no asm body for this MLP shape");')
    out.append('}')
    out.append('')
    out += phase_checks()
    for kind in ('ALL', 'REST', 'FORCE2'):
        out += gen_phase(kind)
    with open(OUT, 'w') as f:
        f.write('\n'.join(out) + '\n')
    print('wrote', OUT, sum(len(l) for l in out), 'bytes')
    for shape in SHAPES:
        print(shape, 'record', record_len(*shape), 'groups', record_len(*shape) // GROUP)


if __name__ == '__main__':
    main()
    gen_dual_file()
    gen_actor_dense()
    gen_actor_mfma_file()
    gen_actor_mfma16_file()
