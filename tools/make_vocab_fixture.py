#!/usr/bin/env python3
"""Writes tests/golden/vocab_subset.json: the vocab.txt lines (id -> token string) of the ids that occur in the
reference's expected_tokens.txt, a few special tokens and non-ASCII examples.  Data fixture; dev container only
(reads /root/reference/vocab.txt, which the reference's export_weights.py:134-142 produced)."""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lines = open("/root/reference/vocab.txt", encoding="utf-8").read().split("\n")
ids = sorted(set(int(x) for x in re.findall(r"\((\d+)\)", open(os.path.join(ROOT, "tests/golden/expected_tokens.txt")).read())))
ids += [0, 1, 220, 256, 257, 50255, 50256, 50257, 50258, 50259, 50359, 50363, 50364]
ids += [i for i, t in enumerate(lines) if t in ("åľº", "Ġcafé", "ĠÃ", "\\n", "\\n\\n")][:6]
out = {str(i): lines[i] for i in sorted(set(ids))}
json.dump({"n_lines": len(lines), "tokens": out}, open(os.path.join(ROOT, "tests/golden/vocab_subset.json"), "w"), ensure_ascii=False, indent=0)
print(len(out), "entries;", len(lines), "split entries")
