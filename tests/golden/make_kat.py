"""Writes tests/golden/kat_reference.json: the known-answer vectors the reference's own tests hold
for the Smith-Waterman path, transcribed as data (inputs + expected aligned strings).

Sources (all under the reference tree, read as text):
  rust/src/lib.rs:271-410   21 vectors, run against find_alignment_sequential,
                            find_alignment_sequential_straight, find_alignment_simd::<64> and
                            find_alignment_simd_lowmem::<64> (lib.rs:423-457); only the two aligned
                            strings are asserted (lib.rs:266-268).
  internal/smithwaterman/smithwaterman_test.go:90   one extra vector (pure-Go scalar SW; only the query
                            string is compared there).
  master/tests/data.py:43-63  protocol fixture: "ABCD" vs "ABCD", match 1 -> score 4, maxX = maxY = 3
                            (confirms the 0-based inclusive end coordinates).
Scores are {gap, match, miss} as the Rust side sees them (penalties negative).
"""
import json
import os

S = dict(gap=-1, match=2, miss=-1)
S3 = dict(gap=-1, match=3, miss=-1)
S5 = dict(gap=-3, match=5, miss=-2)
LONG_Q = "x" * 1000 + "abc" + "x" * 500
vec = [
    ("A", "A", "A", "A", S), ("HOI", "HOI", "HOI", "HOI", S),
    ("AAAAAAATAAAAAAAA", "CCTCCCCCCCCCCCCC", "T", "T", S),
    ("A", "T", "", "", S), ("AAAA", "TTTT", "", "", S),
    ("ATATTTATTAAATATATTATATATTAA", "CCCCGCGGGGCGCGCGGCGCGCGCGCGCG", "", "", S),
    ("CCAA", "GATA", "A-A", "ATA", S), ("AA", "ATA", "A-A", "ATA", S), ("AA", "ATTA", "A", "A", S),
    ("AAAAAAAAA", "AAATTAAATTAAA", "AAA--AAA--AAA", "AAATTAAATTAAA", S),
    ("AA", "ATTA", "A--A", "ATTA", S3), ("ATA", "ATTA", "A-TA", "ATTA", S3),
    ("ATA", "ACA", "ATA", "ACA", S),
    ("ACAC", "ACGCTTTTACC", "ACAC", "ACGC", S5), ("ACAC", "AGGCTTTTACC", "ACAC", "AC-C", S5),
    ("AA", "AATAA", "AA", "AA", S), ("ATTA", "ATAA", "ATTA", "A-TA", S),
    ("TACGGGCCCGCTAC", "TAGCCCTATCGGTCA", "TACGGGCCCGCTA-C", "TA---G-CC-CTATC", S),
    ("AAGTCGTAAAAGTGCACGT", "TAAGCCGTTAAGTGCGCGTG", "AAGTCGTAAAAGTGCACGT", "AAGCCGT-TAAGTGCGCGT", S),
    (LONG_Q, "z" * 2000 + "abc" + "z" * 1000, "abc", "abc", S),
    (LONG_Q, "z" * 2000 + "ac" + "z" * 1000, "abc", "a-c", S),
]
out = {
    "rust_lib_rs": [dict(id=i + 1, query=q, target=t, q_aln=qa, t_aln=ta, scores=s)
                    for i, (q, t, qa, ta, s) in enumerate(vec)],
    "go_smithwaterman_test": [dict(query="AAGTCGTAAAAGTGCACGT", target="z" * 230 + "TAAGCCGTTAAGTGCGCGTG",
                                   q_aln="AAGTCGTAAAAGTGCACGT", scores=S)],
    "master_tests_data": [dict(query="ABCD", target="ABCD", q_aln="ABCD", t_aln="ABCD", score=4, max_x=3,
                               max_y=3, scores=dict(gap=-1, match=1, miss=-1))],
}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_reference.json"), "w") as f:
    json.dump(out, f, indent=1)
