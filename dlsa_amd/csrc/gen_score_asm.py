#!/usr/bin/env python3
"""Generates dlsa_amd/csrc/sw_score_asm.inc: the hand-scheduled gfx950 row loop of the inter-task score
kernel (sw_score.hip) as one inline-asm string.

Why generated asm: the loop body exists in 8 variants (4 query nucleotides x 2 row parities) that must all
update the 16 H registers IN PLACE and read 64 table registers; hipcc's register allocator inserted 16-32
register copies per row around any C++ / per-cell inline-asm formulation (see DESIGN.md "Score kernel").
Here every register is named explicitly:

  v48..v63    H[0..15]            (pinned in/out operands of the asm statement, 4 tuples of 4)
  v64..v79    tabA[0..15]   v80..v95 tabC   v96..v111 tabG   v112..v127 tabT   (pinned inputs)
  v40 m  v41 n  v42 a  v43 d  v44 L1  v45 D1  v46 L0  v47 D0     (clobbered temporaries)

Per cell (see sw_score.hip for the recurrence): v_pk_max_i16, v_pk_sub_u16 clamp, v_pk_add_i16 clamp (the
NEXT cell's diag + s, computed before the old H is overwritten), v_pk_max_i16, and on odd columns the running
maximum.  gfx950 needs one wait state between a packed op and a dependent packed op; every such pair is
separated by an independent instruction or an s_nop.

Row loop, unrolled by two so that the left/diag values alternate between (L0, D0) and (L1, D1) without moves:
the left neighbour's column arrives through LDS at [addr + 256 * row]; the value a row needs as `left` is also
the next row's `diag`, and is simply read twice.  The wave's own last column leaves through [waddr + 256 * row]
(the right neighbour's ring) and, when bit 9 of [n] is set, also through global memory.
"""
import os
import sys

H0, TAB0 = 48, 64
M, N, A, D, L1, D1, L0, D0 = 40, 41, 42, 43, 44, 45, 46, 47


def h(r):
    return f"v{H0 + r}"


def tab(code, r):
    return f"v{TAB0 + 16 * code + r}"


def cells4(code, r0, hl):
    """four cells r0..r0+3; d (v43) holds diag+s of cell r0 on entry and of cell r0+4 on exit"""
    t = lambda r: tab(code, min(r, 15))
    h0, h1, h2, h3 = h(r0), h(r0 + 1), h(r0 + 2), h(r0 + 3)
    m, n, a, d = f"v{M}", f"v{N}", f"v{A}", f"v{D}"
    return [
        f"v_pk_max_i16 {m}, {hl}, {h0}",
        f"v_pk_add_i16 {a}, {h0}, {t(r0 + 1)} clamp",
        f"v_pk_sub_u16 {m}, {m}, %[g] clamp",
        "s_nop 0",
        f"v_pk_max_i16 {h0}, {m}, {d}",
        "s_nop 0",
        f"v_pk_max_i16 {m}, {h0}, {h1}",
        f"v_pk_add_i16 {d}, {h1}, {t(r0 + 2)} clamp",
        f"v_pk_sub_u16 {n}, {m}, %[g] clamp",
        f"v_pk_max_i16 %[best], %[best], {m}",
        f"v_pk_max_i16 {h1}, {n}, {a}",
        "s_nop 0",
        f"v_pk_max_i16 {m}, {h1}, {h2}",
        f"v_pk_add_i16 {a}, {h2}, {t(r0 + 3)} clamp",
        f"v_pk_sub_u16 {m}, {m}, %[g] clamp",
        "s_nop 0",
        f"v_pk_max_i16 {h2}, {m}, {d}",
        "s_nop 0",
        f"v_pk_max_i16 {m}, {h2}, {h3}",
    ] + ([f"v_pk_add_i16 {d}, {h3}, {t(r0 + 4)} clamp"] if r0 + 4 < 16 else ["s_nop 0"]) + [
        f"v_pk_sub_u16 {n}, {m}, %[g] clamp",
        f"v_pk_max_i16 %[best], %[best], {m}",
        f"v_pk_max_i16 {h3}, {n}, {a}",
        "s_nop 0",
    ]


def body(code, parity):
    left, diag = (f"v{L0}", f"v{D0}") if parity == 0 else (f"v{L1}", f"v{D1}")
    out = [f"v_pk_add_i16 v{D}, {diag}, {tab(code, 0)} clamp"]
    out += cells4(code, 0, left)
    for r0 in (4, 8, 12):
        out += cells4(code, r0, h(r0 - 1))
    return out


def row(parity):
    p = "e" if parity == 0 else "o"
    nl, nd = (L1, D1) if parity == 0 else (L0, D0)
    off = 256 * parity
    # LDS ops retire in order.  Outstanding here, oldest first: this row's two reads (issued one row ago), the
    # previous row's write, and the two prefetches just issued -> lgkmcnt(3) waits for exactly the two reads
    # this row needs and lets the write and the prefetches fly.  Every wave issues exactly one write per row
    # (a wave without a right-hand neighbour writes into an unused ring) so that the count is static.
    o = [f"ds_read_b32 v{nl}, %[addr] offset:{off + 256}",   # next row's left
         f"ds_read_b32 v{nd}, %[addr] offset:{off}",         # this row's left = next row's diag
         "s_waitcnt lgkmcnt(3)",
         "s_bitcmp1_b32 %[qw], 1", f"s_cbranch_scc1 Lhi_{p}_%=",
         "s_bitcmp1_b32 %[qw], 0", f"s_cbranch_scc1 LC_{p}_%="]
    o += body(0, parity) + [f"s_branch Ljoin_{p}_%=", f"LC_{p}_%=:"]
    o += body(1, parity) + [f"s_branch Ljoin_{p}_%=", f"Lhi_{p}_%=:",
                            "s_bitcmp1_b32 %[qw], 0", f"s_cbranch_scc1 LT_{p}_%="]
    o += body(2, parity) + [f"s_branch Ljoin_{p}_%=", f"LT_{p}_%=:"]
    o += body(3, parity) + [f"Ljoin_{p}_%=:", "s_lshr_b32 %[qw], %[qw], 2",
                            f"ds_write_b32 %[waddr], {h(15)} offset:{off}",
                            "s_bitcmp1_b32 %[n], 9", f"s_cbranch_scc0 Lng_{p}_%=",
                            f"global_store_dword %[goff], {h(15)}, %[gb] sc1",
                            "v_add_u32_e32 %[goff], 0x100, %[goff]", f"Lng_{p}_%=:"]
    return o


def generate():
    o = [f"v_mov_b32_e32 v{D0}, %[diag]",
         f"ds_read_b32 v{L0}, %[addr]",
         "s_and_b32 %[i], %[n], 0xff",
         "s_waitcnt lgkmcnt(0)",
         "Ltop_%=:"]
    o += row(0)
    o += ["s_sub_u32 %[i], %[i], 1", "s_cmp_eq_u32 %[i], 0", "s_cbranch_scc1 Lexit_%="]
    o += row(1)
    o += ["v_add_u32_e32 %[addr], 0x200, %[addr]",
          "v_add_u32_e32 %[waddr], 0x200, %[waddr]",
          "s_sub_u32 %[i], %[i], 1", "s_cmp_lg_u32 %[i], 0", "s_cbranch_scc1 Ltop_%=",
          "Lexit_%=:",
          "s_waitcnt lgkmcnt(0)",   # the prefetches into L0/D0 have landed before the registers are released
          f"v_mov_b32_e32 %[diag], v{D0}"]
    return o


def is_valu(ln):
    return ln.startswith("v_")


def counts():
    """Instruction counts of ONE query row of one strip (16 columns x 2 targets per lane), read off the generated
    text so that measurement code (sw_hip_last_stats, bench.py) never carries a hand-copied constant: packed VALU
    instructions of a row body, the other VALU instructions of the loop per row (address increments), s_nops."""
    b = body(0, 0)
    assert all(len(body(c, p)) == len(b) for c in range(4) for p in range(2))
    packed = sum(1 for ln in b if ln.startswith("v_pk_"))
    snop = sum(1 for ln in b if ln.startswith("s_nop"))
    loop = generate()
    top = loop.index("Ltop_%=:")
    two_rows = loop[top:loop.index("Lexit_%=:")]
    other = sum(1 for ln in two_rows if is_valu(ln) and not ln.startswith("v_pk_"))
    # the loop text holds 4 bodies per row parity; `other` counts the per-2-row address increments plus one
    # v_add per row parity that only executes when the row is also stored to global memory
    aux_always = sum(1 for ln in two_rows if ln.startswith("v_add_u32_e32 %[addr]") or ln.startswith("v_add_u32_e32 %[waddr]"))
    aux_store = sum(1 for ln in two_rows if ln.startswith("v_add_u32_e32 %[goff]"))
    assert other == aux_always + aux_store and aux_always == 2 and aux_store == 2
    return {"packed": packed, "snop": snop, "aux_x2": aux_always, "store": aux_store // 2}


def render(Dslots=3, B=12):
    """the text of sw_score_asm.inc (ring geometry must match sw_score.hip: D slots x B rows x 256 bytes per wave;
    the asm text itself does not depend on them; B must be even)"""
    lines = generate()
    c = counts()
    out = ["// GENERATED by gen_score_asm.py -- do not edit.  Row loop of sw_score_kernel (gfx950).\n",
           f"#define SW_SCORE_ASM_D {Dslots}\n#define SW_SCORE_ASM_B {B}\n",
           "// executed instructions per query row of one strip (wave-instructions; 16 columns x 128 targets):\n",
           f"#define SW_SCORE_ASM_PACKED_PER_ROW {c['packed']}   /* v_pk_* of one row body */\n",
           f"#define SW_SCORE_ASM_AUX_PER_2ROWS {c['aux_x2']}    /* address increments, once per two rows */\n",
           f"#define SW_SCORE_ASM_STORE_PER_ROW {c['store']}    /* extra VALU of a row that is also stored to HBM */\n",
           f"#define SW_SCORE_ASM_SNOP_PER_ROW {c['snop']}\n",
           "#define SW_SCORE_ROWS_ASM \\\n"]
    for ln in lines:
        out.append(f'  "{ln}\\n\\t" \\\n')
    out.append('  ""\n')
    return "".join(out), len(lines)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    text, n = render()
    if "--check" in sys.argv:  # used by build.py and tests/test_cpu_generated_asm.py
        with open(os.path.join(here, "sw_score_asm.inc")) as f:
            if f.read() != text:
                print("sw_score_asm.inc is stale: run python dlsa_amd/csrc/gen_score_asm.py", file=sys.stderr)
                sys.exit(1)
        return
    with open(os.path.join(here, "sw_score_asm.inc"), "w") as f:
        f.write(text)
    print(n, "asm lines")


if __name__ == "__main__":
    main()
