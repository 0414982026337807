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


base.row = row
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
