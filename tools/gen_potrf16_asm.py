#!/usr/bin/env python3
"""Generates tools/micro/potrf16_asm.inc (an EXPERIMENT, not part of the product, see the end of this note): the in-register 16x16 Cholesky + inverse of chol.hip
(wave_potrf16) as ONE inline-asm block with a hand-made instruction order.

Why generated assembly: a wave that runs alone on its SIMD pays ~5.3 cycles for an independent fp64 VALU instruction,
~8.4 for one that needs the previous result, ~8.5 for every s_nop, ~20 for v_rsq_f64 (tools/micro/fp64_issue.hip).  The
C++ form spends five s_nop per pivot on hazards (VALU write -> DPP read: 2 wait states; transcendental result -> VALU
read: 1) and keeps the 16 x (6-instruction reciprocal-root chain) strictly behind the column updates.  Here
  * the chain of pivot J+1 is interleaved with the independent column updates of pivot J,
  * every hazard slot holds useful work (the stores of the finished columns J-1 go there too),
  * an s_nop is emitted only where the checker below finds a hazard with nothing left to put in front of it.
The arithmetic (operations, operands, order of roundings per value) is exactly that of the C++ form, so results are
bit-identical; tools/micro/potrf16.hip checks that and times both.

Register plan (fixed VGPRs, declared as clobbers): a[c] = v[2c:2c+1], w[c] in v32-39/48-55/64-71/80-87, temporaries v96-119.
Operands: %0 = ylast (out), %1 = LDS byte address of this lane's row of the block (columns at +8c),
%2 = LDS byte address of this lane's row of the identity (+8c), %3 = LDS byte address Wl + 8*row (column c at +128c).
Runs with EXEC = lanes 0..15 (the caller's `if (lane < 16)`).

Outcome (round 2): in isolation the block drops from 3744 to 3282 cycles (tools/micro/potrf16.hip, variant 5 vs 0) and in
the factorisation kernel the stamped pivot block from 1.56 to 1.42 us -- but the whole factorisation got 5 us SLOWER
(294.8 -> 299.7 us, A/B of the default bench), as did a build whose pivot wave skipped the identity rows altogether
(pivot block 1.36 us, factorisation 305 us).  The pivot block is not what bounds a block column any more; the seven
dependent 16x16x16 fp64 GEMM hops per block column (0.45-1.0 us each: operand loads, four dependent 64-cycle MFMAs, store)
and the strip flight are.  The shipped kernel keeps the C++ form.
"""
import os
import sys

# Only registers the AMDGPU calling convention lets a callee clobber (v0-v39, v48-55, v64-71, v80-87, v96-103, v112-119):
# wave_potrf16 is a noinline function, so its caller keeps live values in the callee-saved groups and neither side spills.
WBASE = [32, 48, 64, 80]            # w columns 4g..4g+3 live in the 8-register group WBASE[g]


def reg(first):
    return f"v[{first}:{first + 1}]"


def a(c):
    return reg(2 * c)


def w(c):
    return reg(WBASE[c // 4] + 2 * (c % 4))


D, Y0, T, E = reg(96), reg(98), reg(100), reg(102)
P, YE, Y, K38 = reg(112), reg(114), reg(116), reg(118)   # K38 = 0.375
CLOBBER = list(range(0, 40)) + list(range(48, 56)) + list(range(64, 72)) + list(range(80, 88)) + list(range(96, 104)) + list(range(112, 120))


class I:
    def __init__(self, text, writes=(), reads=(), dpp=(), trans=False):
        self.text, self.writes, self.reads, self.dpp, self.trans = text, set(writes), set(reads), set(dpp), trans


def fmac(dst, piv, own, lane):
    return I(f"v_fmac_f64_dpp {dst}, -{piv}, {own} row_newbcast:{lane} row_mask:0xf bank_mask:0xf",
             writes=[dst], reads=[dst, own, piv], dpp=[piv])


def chain(j):
    """1/sqrt(pivot j) -> Y: the instructions of rsqrt_nr (chol.hip), in its order."""
    return [
        I(f"v_mov_b64_dpp {D}, {a(j)} row_newbcast:{j} row_mask:0xf bank_mask:0xf", writes=[D], reads=[a(j)], dpp=[a(j)]),
        I(f"v_rsq_f64 {Y0}, {D}", writes=[Y0], reads=[D], trans=True),
        I(f"v_mul_f64 {T}, {Y0}, -{D}", writes=[T], reads=[Y0, D]),              # -d*y
        I(f"v_fma_f64 {E}, {T}, {Y0}, 1.0", writes=[E], reads=[T, Y0]),          # e = fma(-d*y, y, 1)
        I(f"v_fma_f64 {P}, {E}, {K38}, 0.5", writes=[P], reads=[E]),             # p = fma(0.375, e, 0.5)
        I(f"v_mul_f64 {YE}, {Y0}, {E}", writes=[YE], reads=[Y0, E]),             # y*e
        I(f"v_fma_f64 {Y}, {YE}, {P}, {Y0}", writes=[Y], reads=[YE, P, Y0]),     # y + (y e) p
    ]


def build():
    prog = []
    prog.append(I("v_mov_b32 v118, 0", writes=[K38]))
    prog.append(I("v_mov_b32 v119, 0x3fd80000", writes=[K38]))
    for k in range(8):   # rows of the block, then of the identity: 16-byte reads
        prog.append(I(f"ds_read_b128 v[{4 * k}:{4 * k + 3}], %1 offset:{16 * k}", writes=[a(2 * k), a(2 * k + 1)]))
    for k in range(8):   # (at most 15 LDS operations outstanding: the counter has four bits)
        if k == 7:
            prog.append(I("s_waitcnt lgkmcnt(7)"))
        wb = WBASE[k // 2] + 4 * (k % 2)
        prog.append(I(f"ds_read_b128 v[{wb}:{wb + 3}], %2 offset:{16 * k}", writes=[w(2 * k), w(2 * k + 1)]))
    prog += chain(0)
    prog.append(I("s_waitcnt lgkmcnt(0)"))
    for j in range(16):
        # scale column j
        head = [I(f"v_mul_f64 {a(j)}, {a(j)}, {Y}", writes=[a(j)], reads=[a(j), Y]),
                I(f"v_mul_f64 {w(j)}, {w(j)}, {Y}", writes=[w(j)], reads=[w(j), Y])]
        if j == 15:
            head.append(I(f"v_mov_b64 %0, {Y}", reads=[Y]))
        stores = []
        if j >= 1:
            stores = [I(f"ds_write_b64 %1, {a(j - 1)} offset:{8 * (j - 1)}", reads=[a(j - 1)]),
                      I(f"ds_write_b64 %3, {w(j - 1)} offset:{128 * (j - 1)}", reads=[w(j - 1)])]
        if j == 15:
            stores += [I(f"ds_write_b64 %1, {a(15)} offset:{8 * 15}", reads=[a(15)]),
                       I(f"ds_write_b64 %3, {w(15)} offset:{128 * 15}", reads=[w(15)])]
        upd = []
        for c in range(j + 1, 16):
            upd.append(fmac(a(c), a(j), a(j), c))
            upd.append(fmac(w(c), a(j), w(j), c))
        nxt = chain(j + 1) if j < 15 else []
        # order: scale, one store (second wait state in front of the first DPP read of a[j]), the next pivot column's
        # update, then the chain of pivot j+1 with the remaining work dealt in between, two fillers per chain step
        seq = list(head)
        fill = []
        if stores:
            seq.append(stores[0])
            fill_first = stores[1:]
        else:
            fill_first = []
        if upd:
            seq.append(upd[0])          # a[j+1]
            rest = upd[1:]
        else:
            rest = []
        fill = rest[:1] + fill_first + rest[1:]   # w[j+1], the second store, a[j+2], ...
        per_gap = 2
        for ci, cinstr in enumerate(nxt):
            take = per_gap if ci > 0 else 2
            if ci == 0:
                seq += fill[:take]
                fill = fill[take:]
                seq.append(cinstr)
            else:
                seq += fill[:take]
                fill = fill[take:]
                seq.append(cinstr)
        seq += fill
        prog += seq
    return prog


def fix_hazards(prog):
    """VALU write -> DPP read of the same VGPR: 2 wait states; transcendental write -> VALU read: 1 wait state."""
    out = []
    nops = 0
    for ins in prog:
        need = 0
        for back in range(1, 3):
            if len(out) >= back:
                prev = out[-back]
                if prev.writes & ins.dpp:
                    need = max(need, 3 - back)          # writer at distance 1 -> 2 nops, distance 2 -> 1
                if back == 1 and prev.trans and (prev.writes & ins.reads):
                    need = max(need, 1)
        for _ in range(need):
            out.append(I("s_nop 0"))
            nops += 1
        out.append(ins)
    return out, nops


def main():
    prog, nops = fix_hazards(build())
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(here, "micro", "potrf16_asm.inc")
    clob = [f"v{i}" for i in CLOBBER] + ["memory"]
    with open(dst, "w") as f:
        f.write("// GENERATED by tools/gen_potrf16_asm.py -- do not edit.  In-register 16x16 Cholesky + inverse, one asm block;\n")
        f.write(f"// {len(prog)} instructions, {nops} of them s_nop.  See the generator for the schedule and the register plan.\n")
        f.write("#define DROID_POTRF16_ASM \\\n")
        for ins in prog:
            f.write(f'  "{ins.text}\\n\\t" \\\n')
        f.write('  ""\n')
        f.write("#define DROID_POTRF16_CLOBBERS " + ", ".join(f'"{c}"' for c in clob) + "\n")
    print(f"wrote {dst}: {len(prog)} instructions, {nops} s_nop", file=sys.stderr)


if __name__ == "__main__":
    main()
