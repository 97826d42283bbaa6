"""Tokenizer mirror: tokenizer.mojo behaviour on the vocab lines the reference ships, and the correct
byte-level decoding next to it.  Fixture: tests/golden/vocab_subset.json (tools/make_vocab_fixture.py)."""
import json
import os
import re

import pytest

from whisper_mojo_amd.tokenizer import Tokenizer

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def vocab():
    d = json.load(open(os.path.join(G, "vocab_subset.json"), encoding="utf-8"))
    return {int(k): v for k, v in d["tokens"].items()}, d["n_lines"]


def test_expected_tokens_decode_to_the_readme_transcript(vocab):
    """expected_tokens.txt (89 ids) renders to the sentence the reference's README quotes (readme.md:75)."""
    tok = Tokenizer(vocab[0])
    ids = [int(x) for x in re.findall(r"\((\d+)\)", open(os.path.join(G, "expected_tokens.txt")).read())]
    text = tok.decode(ids)
    assert text.startswith(" This is my voice on the left.") and text.endswith("My voice will be out of phase on three.")
    assert tok.decode_text(ids) == text  # pure ASCII: both renderings agree
    # what Whisper.transcribe returns = prompt + ids + eot: special tokens are filtered (tokenizer.mojo:22)
    assert tok.decode([50258, 50259, 50359, 50363] + ids + [50257]) == text


def test_reference_rendering_mangles_non_ascii_and_byte_level_decoding_fixes_it(vocab):
    tok = Tokenizer(vocab[0])
    assert vocab[0][50255] == "åľº"
    assert tok.decode([50255]) == "åľº"          # bug-compatible with tokenizer.mojo:24
    assert tok.decode_text([50255]) == "场"      # e5 9c ba
    assert tok.decode([220]) == " " and tok.decode_text([220]) == " "


def test_file_loading_matches_reference_split(tmp_path, vocab):
    lines = ["!", "Ġthe", "<|endoftext|>", "a\\\\nb".replace("\\\\", "\\"), "ĠcafÃ©"]
    p = tmp_path / "vocab.txt"
    p.write_text("\n".join(lines) + "\n", encoding="utf-8")
    tok = Tokenizer(str(p))
    assert len(tok.vocab) == len(lines) + 1          # trailing newline -> one extra empty entry (tokenizer.mojo:11)
    assert vocab[1] == 51866                          # the real file: 51 865 lines -> 51 866 split entries
    assert tok.decode([1, 0, 2, 3, 99, -1]) == " the!a\nb"   # out-of-range ids are skipped (tokenizer.mojo:19)
    assert tok.decode_text([4]) == " café"
    with pytest.raises(OSError):
        Tokenizer(str(tmp_path / "missing.txt"))
