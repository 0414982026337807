#!/usr/bin/env python3
"""Generates tests/golden/wire_golden.json IN THE BUILD CONTAINER, where /root/reference is mounted:

  * "fasta": tricky FASTA inputs and what the reference's own parser (cli/__main__.py:19-34 parse_fasta, imported
    from the reference tree) returns for them -- the golden vectors of sw_fasta_parse;
  * "work_result_schema" / "work_package_schema": the JSON schemas of the master's pydantic models for the body of
    POST /work/{id}/result and of a work package (master/api_models/work.py:8-41, imported from the reference tree)
    -- the golden vectors of dlsa_amd.worker's wire structs.

Only data is written (inputs, outputs, schemas); nothing of the reference's source travels.  The tests read the
JSON and never touch /root/reference.
"""
import json
import os
import sys
import tempfile

REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

FASTA_CASES = {
    "plain": ">seq0\nACGT\n>seq1\nTTGA\n",
    "multiline_body": ">q1 some description here\nACGT\nGGCC\nTT\n>q2\tdesc\nAAAA\n\nCCCC\n",
    "crlf": ">a\r\nACGT\r\nGG\r\n>b\r\nTT\r\n",
    "lone_cr": ">a\rACGT\rGG\r>b\rTT",
    "no_trailing_newline": ">only\nACGTACGT",
    "leading_junk_and_blank_lines": "junk before\n\n>x\n\nAC\n\n\nGT\n\n",
    "inner_whitespace_kept": ">w\nAC GT\n  TTAA  \n\tCC\n",
    "gt_inside_line": ">a\nAC>b GT\nTT\n",
    "lowercase_and_iupac": ">n1\nacgtnnryACGT\n>n2\nNNNN\n",
    "empty_sequence": ">e1\n>e2\nAC\n",
    "header_with_spaces_first": ">   id_after_spaces rest\nACGT\n",
    "unicode": ">u1 café\nACéGT\n",
    "header_only_is_its_own_id": ">\nACGT\n",
    "record_of_whitespace_only": ">a\nAC\n>  \n>b\nGT\n",
    "empty_file": "",
    "no_records": "ACGT\nTTTT\n",
}


def main():
    import cli.__main__ as ref_cli  # the reference's own parser
    from master.api_models import work as ref_work

    fasta = []
    for name, text in FASTA_CASES.items():
        with tempfile.NamedTemporaryFile("w", suffix=".fasta", delete=False, newline="", encoding="utf-8") as f:
            f.write(text)
            path = f.name
        try:
            ref_cli.descr_map.clear()
            recs = ref_cli.parse_fasta(path)
            ids = [ref_cli.descr_map[u] for u, _ in recs]
            expected = {"ids": ids, "seqs": [s for _, s in recs]}
        except Exception as e:  # e.g. IndexError on a record without an id
            expected = {"error": type(e).__name__}
        finally:
            os.unlink(path)
        fasta.append({"name": name, "text": text, "expected": expected})

    out = {
        "fasta": fasta,
        "work_result_schema": ref_work.WorkResult.model_json_schema(),
        "work_package_schema": ref_work.WorkPackage.model_json_schema(),
    }
    with open(os.path.join(HERE, "wire_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote wire_golden.json:", len(fasta), "fasta cases")


if __name__ == "__main__":
    main()
