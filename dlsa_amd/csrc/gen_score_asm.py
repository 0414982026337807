#!/usr/bin/env python3
"""Generates dlsa_amd/csrc/sw_score_asm.inc: the hand-scheduled gfx950 row loop of the inter-task score
kernel (sw_score.hip) as one inline-asm string.

Why generated asm: the loop body must update the 16 H registers IN PLACE and read one of four tables of 16
registers, chosen by the row's query nucleotide; hipcc's register allocator inserted 16-32 register copies per
row around any C++ / per-cell inline-asm formulation (see DESIGN.md "Score kernel").  Here every register is named
explicitly:

  v48..v63    H[0..15]            (pinned in/out operands of the asm statement, 4 tuples of 4)
  v64..v127   tab[column][nucleotide]: v(64 + 4 r + c) = score of column r against nucleotide c (A, C, G, T)
              (pinned inputs, one tuple of 4 per column)
  v40 m  v41 n  v42 a  v43 d  v44 L1  v45 D1  v46 L0  v47 D0     (clobbered temporaries)

The table is selected WITHOUT a branch: the loop runs in VGPR index mode (s_set_gpr_idx_on ... gpr_idx(SRC2)), the
row's nucleotide code goes into the index with one s_set_gpr_idx_idx, and diag + s is a v_pk_mad_i16 h * 1 + tab
whose third source names the A column register -- the hardware adds the index, so the same instruction reads the
A, C, G or T entry.  Only that instruction has a third source, so nothing else is affected.  (The first version had
four copies of the body behind a 4-way scalar branch: 2 % slower on the bench shape, 4 x the code.)  M0 holds the
index and is saved / restored around the loop (the compiler keeps the LDS-DMA base in it).

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


def tab(r):
    """the A entry of column r; index mode adds the row's nucleotide code (0..3)"""
    return f"v{TAB0 + 4 * r}"


def mad(dst, hsrc, r):
    """dst = sat(hsrc + tab[r][nucleotide of this row]) for both packed halves"""
    return f"v_pk_mad_i16 {dst}, {hsrc}, %[one], {tab(r)} clamp"


def cells4(r0, hl, cols=16):
    """four cells r0..r0+3; d (v43) holds diag+s of cell r0 on entry and of cell r0+4 on exit"""
    h0, h1, h2, h3 = h(r0), h(r0 + 1), h(r0 + 2), h(r0 + 3)
    m, n, a, d = f"v{M}", f"v{N}", f"v{A}", f"v{D}"
    return [
        f"v_pk_max_i16 {m}, {hl}, {h0}",
        mad(a, h0, r0 + 1),
        f"v_pk_sub_u16 {m}, {m}, %[g] clamp",
        "s_nop 0",
        f"v_pk_max_i16 {h0}, {m}, {d}",
        "s_nop 0",
        f"v_pk_max_i16 {m}, {h0}, {h1}",
        mad(d, h1, r0 + 2),
        f"v_pk_sub_u16 {n}, {m}, %[g] clamp",
        f"v_pk_max_i16 %[best], %[best], {m}",
        f"v_pk_max_i16 {h1}, {n}, {a}",
        "s_nop 0",
        f"v_pk_max_i16 {m}, {h1}, {h2}",
        mad(a, h2, r0 + 3),
        f"v_pk_sub_u16 {m}, {m}, %[g] clamp",
        "s_nop 0",
        f"v_pk_max_i16 {h2}, {m}, {d}",
        "s_nop 0",
        f"v_pk_max_i16 {m}, {h2}, {h3}",
    ] + ([mad(d, h3, r0 + 4)] if r0 + 4 < cols else ["s_nop 0"]) + [
        f"v_pk_sub_u16 {n}, {m}, %[g] clamp",
        f"v_pk_max_i16 %[best], %[best], {m}",
        f"v_pk_max_i16 {h3}, {n}, {a}",
        "s_nop 0",
    ]


def body(parity, cols=16):
    """one query row over the first `cols` columns of the strip (16, or 8 for the half-width form that a group's
    last strip takes when its targets end within its first eight columns)"""
    assert cols in (8, 16)
    left, diag = (f"v{L0}", f"v{D0}") if parity == 0 else (f"v{L1}", f"v{D1}")
    out = [mad(f"v{D}", diag, 0)]
    out += cells4(0, left, cols)
    for r0 in range(4, cols, 4):
        out += cells4(r0, h(r0 - 1), cols)
    return out


def row(parity, store, cols=16):
    nl, nd = (L1, D1) if parity == 0 else (L0, D0)
    off = 256 * parity
    # LDS ops retire in order.  Outstanding here, oldest first: this row's two reads (issued one row ago), the
    # previous row's write, and the two prefetches just issued -> lgkmcnt(3) waits for exactly the two reads
    # this row needs and lets the write and the prefetches fly.  Every wave issues exactly one write per row
    # (a wave without a right-hand neighbour writes into an unused ring) so that the count is static.
    o = [f"ds_read_b32 v{nl}, %[addr] offset:{off + 256}",   # next row's left
         f"ds_read_b32 v{nd}, %[addr] offset:{off}",         # this row's left = next row's diag
         "s_waitcnt lgkmcnt(3)",
         "s_and_b32 %[t], %[qw], 3",        # this row's nucleotide selects the table entry (index mode, SRC2)
         "s_set_gpr_idx_idx %[t]",
         "s_nop 0"]
    o += body(parity, cols) + ["s_lshr_b32 %[qw], %[qw], 2",
                               f"ds_write_b32 %[waddr], {h(cols - 1)} offset:{off}"]
    if store:  # the wave whose last column feeds the next unit (another workgroup) through HBM
        assert cols == 16
        o += [f"global_store_dword %[goff], {h(15)}, %[gb] sc1",
              "v_add_u32_e32 %[goff], 0x100, %[goff]"]
    return o


def generate(store=False, cols=16):
    """the loop in one of its forms: without / with the store of the wave's last column to global memory.
    The kernel picks the form per block outside the loop (one wave in 16 stores, and only in chained units); a
    test inside the loop cost every wave a scalar compare and a taken branch per row."""
    o = ["s_mov_b32 %[m0s], m0",                       # the compiler's M0 (LDS-DMA base) comes back at the exit
         f"v_mov_b32_e32 v{D0}, %[diag]",
         f"ds_read_b32 v{L0}, %[addr]",
         "s_and_b32 %[i], %[n], 0xff",
         "s_mov_b32 %[t], 0",
         "s_set_gpr_idx_on %[t], gpr_idx(SRC2)",       # from here to Lexit a VALU third source is v[n + index]
         "s_waitcnt lgkmcnt(0)",
         "Ltop_%=:"]
    o += row(0, store, cols)
    o += ["s_sub_u32 %[i], %[i], 1", "s_cmp_eq_u32 %[i], 0", "s_cbranch_scc1 Lexit_%="]
    o += row(1, store, cols)
    o += ["v_add_u32_e32 %[addr], 0x200, %[addr]",
          "v_add_u32_e32 %[waddr], 0x200, %[waddr]",
          "s_sub_u32 %[i], %[i], 1", "s_cmp_lg_u32 %[i], 0", "s_cbranch_scc1 Ltop_%=",
          "Lexit_%=:",
          "s_set_gpr_idx_off",
          "s_mov_b32 m0, %[m0s]",
          "s_waitcnt lgkmcnt(0)",   # the prefetches into L0/D0 have landed before the registers are released
          f"v_mov_b32_e32 %[diag], v{D0}"]
    return o


def generate_block(store, rows=12, cols=16):
    """a FULL block of `rows` query rows, unrolled: no loop counter, no taken branch, the row's nucleotide extracted
    with one s_bfe, every LDS / global offset an immediate (the addresses move once per block, in the kernel).
    Same registers and the same entry / exit contract as generate(); only blocks shorter than `rows` (the last one
    of a query) go through the loop form."""
    assert rows % 2 == 0 and 256 * (rows + 1) < 65536 and 256 * (rows - 1) < 4096
    o = ["s_mov_b32 %[m0s], m0",
         f"v_mov_b32_e32 v{D0}, %[diag]",
         f"ds_read_b32 v{L0}, %[addr]",
         "s_mov_b32 %[t], 0",
         "s_set_gpr_idx_on %[t], gpr_idx(SRC2)"]
    def index_of_row(i):
        return [f"s_bfe_u32 %[t], %[qw], {hex((2 << 16) | (2 * i))}", "s_set_gpr_idx_idx %[t]"]

    o += index_of_row(0)
    for i in range(rows):
        parity = i & 1
        nl, nd = (L1, D1) if parity == 0 else (L0, D0)
        # LDS ops retire in order; outstanding here, oldest first: this row's two reads (issued inside the previous
        # row's body) and the previous row's write -> lgkmcnt(1) waits for the reads and lets the write fly
        o += ["s_waitcnt lgkmcnt(1)" if i else "s_waitcnt lgkmcnt(0)"]
        b = body(parity, cols)
        # the prefetch of the next row's inputs (its left column entry, and this row's left = its diag) takes the
        # body's first two wait slots
        nops = [k for k, ln in enumerate(b) if ln == "s_nop 0"]
        b[nops[0]] = f"ds_read_b32 v{nl}, %[addr] offset:{256 * (i + 1)}"
        b[nops[1]] = f"ds_read_b32 v{nd}, %[addr] offset:{256 * i}"
        if i + 1 < rows:
            # the NEXT row's index goes where this row's last two wait states were (both lie behind the row's last
            # indexed read, and any instruction serves as a wait state): two issue slots less per row
            last_mad = max(k for k, ln in enumerate(b) if ln.startswith("v_pk_mad_i16"))
            nops = [k for k, ln in enumerate(b) if ln == "s_nop 0" and k > last_mad]
            assert len(nops) >= 2
            nxt = index_of_row(i + 1)
            b[nops[-2]], b[nops[-1]] = nxt[0], nxt[1]
        o += b
        o += [f"ds_write_b32 %[waddr], {h(cols - 1)} offset:{256 * i}"]
        if store:
            assert cols == 16
            o += [f"global_store_dword %[goff], {h(15)}, %[gb] offset:{256 * i} sc1"]
    if store:
        o += [f"v_add_u32_e32 %[goff], {hex(256 * rows)}, %[goff]"]
    o += ["s_set_gpr_idx_off",
          "s_mov_b32 m0, %[m0s]",
          "s_waitcnt lgkmcnt(0)",
          f"v_mov_b32_e32 %[diag], v{D0}"]
    return o


def is_valu(ln):
    return ln.startswith("v_")


def counts():
    """Instruction counts of ONE query row of one strip (16 columns x 2 targets per lane) in the form that does the
    work -- the unrolled full block -- read off the generated text so that measurement code (sw_hip_last_stats,
    bench.py) never carries a hand-copied constant: packed VALU instructions of a row, other VALU instructions per
    two rows and per stored row (none since the block is unrolled: the addresses are immediates; what is left is two
    v_mov per BLOCK of 12 rows and one v_add per stored block, not counted), s_nops."""
    b = body(0)
    assert len(body(1)) == len(b)
    packed = sum(1 for ln in b if ln.startswith("v_pk_"))
    snop = sum(1 for ln in b if ln.startswith("s_nop"))
    blk = generate_block(True)
    other = sum(1 for ln in blk if is_valu(ln) and not ln.startswith("v_pk_"))
    assert sum(1 for ln in blk if ln.startswith("v_pk_")) == 12 * packed and other == 3
    half = sum(1 for ln in body(0, 8) if ln.startswith("v_pk_"))
    assert 2 * half == packed and len(body(1, 8)) == len(body(0, 8))
    return {"packed": packed, "packed_half": half, "snop": snop, "aux_x2": 0, "store": 0}


def render(Dslots=3, B=12, Bsmall=6):
    """the text of sw_score_asm.inc (ring geometry must match sw_score.hip: D slots x B rows x 256 bytes per wave;
    the asm text itself does not depend on them; B must be even).  The unrolled block forms come twice: for blocks
    of B rows (the throughput configuration) and, with the suffix _S, for blocks of Bsmall rows -- a small database
    is a few chains of strips, and a chain advances one block at a time (sw_score.hip: score_block_rows)."""
    c = counts()
    out = ["// GENERATED by gen_score_asm.py -- do not edit.  Row loop of sw_score_kernel (gfx950).\n",
           f"#define SW_SCORE_ASM_D {Dslots}\n#define SW_SCORE_ASM_B {B}\n#define SW_SCORE_ASM_B_SMALL {Bsmall}\n",
           "// executed instructions per query row of one strip (wave-instructions; 16 columns x 128 targets):\n",
           f"#define SW_SCORE_ASM_PACKED_PER_ROW {c['packed']}   /* v_pk_* of one row body */\n",
           f"#define SW_SCORE_ASM_AUX_PER_2ROWS {c['aux_x2']}    /* other VALU per two rows (none: unrolled block, immediate offsets) */\n",
           f"#define SW_SCORE_ASM_STORE_PER_ROW {c['store']}    /* extra VALU of a row that is also stored to HBM (none; one per stored block) */\n",
           f"#define SW_SCORE_ASM_SNOP_PER_ROW {c['snop']}\n",
           f"#define SW_SCORE_ASM_PACKED_PER_HALF_ROW {c['packed_half']}   /* ... of the half-width forms (8 columns) */\n",
           ]
    n = 0
    for name, store, full, cols in (("SW_SCORE_ROWS_ASM", False, 0, 16), ("SW_SCORE_ROWS_STORE_ASM", True, 0, 16),
                                    ("SW_SCORE_BLOCK_ASM", False, B, 16), ("SW_SCORE_BLOCK_STORE_ASM", True, B, 16),
                                    ("SW_SCORE_ROWS_HALF_ASM", False, 0, 8), ("SW_SCORE_BLOCK_HALF_ASM", False, B, 8),
                                    ("SW_SCORE_BLOCK_ASM_S", False, Bsmall, 16), ("SW_SCORE_BLOCK_STORE_ASM_S", True, Bsmall, 16),
                                    ("SW_SCORE_BLOCK_HALF_ASM_S", False, Bsmall, 8)):
        lines = generate_block(store, full, cols) if full else generate(store, cols)
        n += len(lines)
        out.append(f"#define {name} \\\n")
        for ln in lines:
            out.append(f'  "{ln}\\n\\t" \\\n')
        out.append('  ""\n')
    return "".join(out), n


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
