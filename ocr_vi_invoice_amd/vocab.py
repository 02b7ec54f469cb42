"""CTC alphabet and tokenizer for the SVTRv2 recogniser.

Mirrors the reference's alphabet rule (model/rec2/vocab.py:5-21) and id layout
(model/rec2/tokenizer.py:4-22): id 0 = CTC blank, id 1 = pad, characters start at
id 2 in code-point order.  The alphabet is *derived* (printable ASCII + the
Vietnamese toned vowels + d-with-stroke + the dong sign) rather than spelled out,
and pinned by a SHA-256 of the resulting string.
"""
from __future__ import annotations

import hashlib
import unicodedata
from typing import Iterable, List, Sequence

# SHA-256 of VOCAB.encode("utf-8") as measured on the reference.
VOCAB_SHA256 = "7be03b4843f51be56eb5d5ec82e7dad67274e6d2935c5b342ab9bdac3119a568"

_BASE_VOWELS = ["a", "ă", "â", "e", "ê", "i", "o", "ô",
                "ơ", "u", "ư", "y"]
# no tone, acute, grave, hook above, tilde, dot below
_TONES = ["", "́", "̀", "̉", "̃", "̣"]


def _build_vocab() -> str:
    chars = {chr(c) for c in range(0x20, 0x7F)}            # printable ASCII incl. space
    for base in _BASE_VOWELS:
        for tone in _TONES:
            low = unicodedata.normalize("NFC", base + tone)
            assert len(low) == 1, (base, tone, low)
            chars.add(low)
            chars.add(low.upper())
    chars.update("đĐ₫")                     # d-stroke (both cases), dong sign
    return "".join(sorted(chars))


VOCAB: str = _build_vocab()
assert hashlib.sha256(VOCAB.encode("utf-8")).hexdigest() == VOCAB_SHA256, "alphabet drifted"


class Tokenizer:
    """id <-> character map with the reference's layout (tokenizer.py:4-22)."""

    blank_id = 0
    pad_id = 1

    def __init__(self, charset: Iterable[str] = VOCAB):
        self.charset: List[str] = sorted(set(charset))
        self.id_to_token = {i + 2: ch for i, ch in enumerate(self.charset)}
        self.token_to_id = {ch: i for i, ch in self.id_to_token.items()}
        self.num_classes = len(self.charset) + 2

    def encode_one(self, text: str) -> List[int]:
        """Characters outside the alphabet are dropped (tokenizer.py:39)."""
        return [self.token_to_id[c] for c in text if c in self.token_to_id]

    def decode(self, token_ids: Sequence[Sequence[int]]) -> List[str]:
        """Drops blank(0) and pad(1); unknown ids are skipped (tokenizer.py:55-79)."""
        if hasattr(token_ids, "tolist"):
            token_ids = token_ids.tolist()
        out = []
        for ids in token_ids:
            out.append("".join(self.id_to_token[i] for i in ids if i in self.id_to_token))
        return out
