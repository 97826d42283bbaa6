"""Mirror of /root/reference/tokenizer.mojo: id -> text through `vocab.txt` (line index = id,
export_weights.py:134-142).

`decode` reproduces the reference exactly (tokenizer.mojo:15-28): special tokens `<|…|>` are dropped, "Ġ" becomes a
space, the escaped "\\n" a newline — which mangles non-ASCII text, because vocab.txt stores GPT-2 *byte-level* symbols
(e.g. line 50 256 "åľº" is the three bytes e5 9c ba = "场").  `decode_text` adds the correct byte-level un-mapping."""
from __future__ import annotations

from typing import Dict, List, Sequence, Union


def _bytes_to_unicode() -> Dict[int, str]:
    """GPT-2's reversible byte <-> printable-unicode table (the published byte-level BPE alphabet)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return {b: chr(c) for b, c in zip(bs, cs)}


_UNICODE_TO_BYTE = {c: b for b, c in _bytes_to_unicode().items()}


class Tokenizer:
    """tokenizer.mojo:4-28"""

    def __init__(self, path_or_vocab: Union[str, Sequence[str], Dict[int, str]]):
        if isinstance(path_or_vocab, str):
            with open(path_or_vocab, "r", encoding="utf-8") as f:  # raises like tokenizer.mojo:9
                self.vocab: Union[List[str], Dict[int, str]] = f.read().split("\n")  # tokenizer.mojo:10-13
        else:
            self.vocab = path_or_vocab if isinstance(path_or_vocab, dict) else list(path_or_vocab)

    def _token(self, token_id: int):
        if isinstance(self.vocab, dict):
            return self.vocab.get(token_id)
        return self.vocab[token_id] if 0 <= token_id < len(self.vocab) else None  # tokenizer.mojo:19

    @staticmethod
    def _is_special(token: str) -> bool:
        return token.startswith("<|") and token.endswith("|>")  # tokenizer.mojo:22

    def decode(self, tokens: Sequence[int]) -> str:
        """The reference's rendering, bug for bug (tokenizer.mojo:15-28)."""
        result = ""
        for token_id in tokens:
            token = self._token(int(token_id))
            if token is None or self._is_special(token):
                continue
            result += token.replace("Ġ", " ").replace("\\n", "\n")
        return result

    def decode_text(self, tokens: Sequence[int]) -> str:
        """Correct byte-level BPE decoding: every vocab symbol maps back to one byte; the byte string is UTF-8."""
        data = bytearray()
        for token_id in tokens:
            token = self._token(int(token_id))
            if token is None or self._is_special(token):
                continue
            token = token.replace("\\n", "Ċ")  # export_weights.py:139 escaped the newline symbol's raw form
            for ch in token:
                b = _UNICODE_TO_BYTE.get(ch)
                if b is None:
                    data.extend(ch.encode("utf-8"))
                else:
                    data.append(b)
        return data.decode("utf-8", errors="replace")
