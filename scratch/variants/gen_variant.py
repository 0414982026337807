#!/usr/bin/env python3
"""Timing experiments on the score kernel's row loop: a patched row loop written next to this file's copy
(make_variant.sh copies it to the work directory as gen_score_asm.py).  VARIANT env: comma-separated flags
  nonop    drop every s_nop of the row bodies (results may be wrong: timing only)
  onebody  no 4-way dispatch on the query symbol (always body A)
  nowait   no s_waitcnt before the row body
  nostore  no global store branch
"""
import importlib.util
import os
import sys

ROOT = os.environ["SW_ROOT"]
spec = importlib.util.spec_from_file_location("base", os.path.join(ROOT, "dlsa_amd/csrc/gen_score_asm.py"))
base = importlib.util.module_from_spec(spec)
spec.loader.exec_module(base)
flags = set(filter(None, os.environ.get("VARIANT", "").split(",")))

orig_cells4, orig_row = base.cells4, base.row
if "nonop" in flags:
    base.cells4 = lambda code, r0, hl: [ln for ln in orig_cells4(code, r0, hl) if not ln.startswith("s_nop")]


def row(parity):
    o = orig_row(parity)
    if "nowait" in flags:
        o = [ln for ln in o if not ln.startswith("s_waitcnt")]
    if "onebody" in flags:
        p = "e" if parity == 0 else "o"
        i0 = o.index("s_bitcmp1_b32 %[qw], 1")
        i1 = o.index(f"Ljoin_{p}_%=:")
        o = o[:i0] + base.body(0, parity) + o[i1:]
    if "nostore" in flags:
        p = "e" if parity == 0 else "o"
        i0 = o.index("s_bitcmp1_b32 %[n], 9")
        i1 = o.index(f"Lng_{p}_%=:")
        o = o[:i0] + o[i1:]
    return o


if "gpridx" in flags:
    # one body per row parity: the substitution table of the row's nucleotide is selected by VGPR index mode
    # (s_set_gpr_idx, SRC2 relative) on a v_pk_mad_i16 h * 1 + tab instead of a 4-way branch to four bodies
    import re
    def cells4_idx(code, r0, hl):
        out = []
        for ln in orig_cells4(0, r0, hl):
            m = re.match(r"v_pk_add_i16 (v\d+), (v\d+), (v\d+) clamp", ln)
            out.append(f"v_pk_mad_i16 {m.group(1)}, {m.group(2)}, %[one], {m.group(3)} clamp" if m else ln)
        return out
    base.cells4 = cells4_idx
    orig_body = base.body
    def body_idx(code, parity):
        out = []
        for ln in orig_body(0, parity):
            m = re.match(r"v_pk_add_i16 (v\d+), (v\d+), (v\d+) clamp", ln)
            out.append(f"v_pk_mad_i16 {m.group(1)}, {m.group(2)}, %[one], {m.group(3)} clamp" if m else ln)
        return out
    base.body = body_idx
    def row_idx(parity):
        o = orig_row(parity)
        p = "e" if parity == 0 else "o"
        i0 = o.index("s_bitcmp1_b32 %[qw], 1")
        i1 = o.index(f"Ljoin_{p}_%=:")
        return o[:i0] + ["s_and_b32 %[t], %[qw], 3", "s_lshl_b32 %[t], %[t], 4", "s_set_gpr_idx_idx %[t]"] + base.body(0, parity) + o[i1:]
    row = row_idx
    _gen0 = base.generate
    def generate_idx():
        o = _gen0()
        top = o.index("Ltop_%=:")
        o[top:top] = ["s_mov_b32 %[t], 0", "s_set_gpr_idx_on %[t], gpr_idx(SRC2)"]
        ex = o.index("Lexit_%=:")
        o[ex + 1:ex + 1] = ["s_set_gpr_idx_off"]
        return o
base.row = row
if "gpridx" in flags:
    base.generate = generate_idx
orig_generate = base.generate
if "storeool" in flags:
    def generate():
        o = orig_generate()
        tail = ["s_branch Lend_%=:".rstrip(":")]
        for p in ("e", "o"):
            i0 = o.index("s_bitcmp1_b32 %[n], 9")
            assert o[i0 + 1] == f"s_cbranch_scc0 Lng_{p}_%=", o[i0 + 1]
            st = o[i0 + 2:i0 + 4]
            assert st[0].startswith("global_store") and st[1].startswith("v_add_u32_e32 %[goff]")
            o[i0:i0 + 4] = ["s_bitcmp1_b32 %[n], 9 ", f"s_cbranch_scc1 Lst_{p}_%="]
            tail += [f"Lst_{p}_%=:"] + st + [f"s_branch Lng_{p}_%="]
        return o + tail + ["Lend_%=:"]
    base.generate = generate
base.counts = lambda: {"packed": 72, "snop": 25, "aux_x2": 2, "store": 1}
text, n = base.render()
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sw_score_asm.inc"), "w") as f:
    f.write(text)
print(n, "asm lines", sorted(flags))
