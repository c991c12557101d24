"""Tokenizer — the reference's byte-level BPE + hard-coded Llama-3 chat template, behaviour for
behaviour (NFAI.Models.Llama3/Tokenizer.cs).  CPU string work, no kernels; needed so that
"identical prompts" produce identical token ids on both backends.

What the reference does, and this mirrors (including its quirks):
* vocabulary and merges come from GGUF metadata `tokenizer.ggml.tokens` / `tokenizer.ggml.merges`
  (Tokenizer.cs:16-58); token strings are compared as UTF-8 byte sequences;
* `Tokenize` always wraps the prompt in a fixed chat template with the system prompt
  "You are a helpful assistant." when `addBos` (first turn) and a shorter continuation template
  otherwise, with `\\r` removed (Tokenizer.cs:68-91);
* special tokens `<|...|>` are cut out with the regex `<\\|[^|>]+?\\|>` and looked up whole; the text
  between them is split with the Llama-3 pre-tokenizer regex (Tokenizer.cs:93-111);
* initial BPE units are one unit per UTF-16 code unit: its UTF-8 bytes, except ' ' and '\\n', which
  become 'Ġ' (C4 A0) and 'Ċ' (C4 8A) (Tokenizer.cs:242-267) — NOT the full GPT-2 byte-to-unicode
  map, so non-ASCII text tokenizes differently from llama.cpp; kept as is;
* merges are applied lowest-rank-first, leftmost on ties (Tokenizer.cs:130-166); a unit that is not
  in the vocabulary raises (Tokenizer.cs:174-177);
* `Detokenize` concatenates token bytes, decodes UTF-8 and maps 'Ġ'->' ', 'Ċ'->'\\n' (Tokenizer.cs:432-462).
"""
from __future__ import annotations

import regex as re

SYSTEM_PROMPT = "You are a helpful assistant."
_SPECIAL = re.compile(r"<\|[^|>]+?\|>", re.IGNORECASE)
_PRETOK = re.compile(
    r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+")


def _utf16_units(s: str):
    """C# strings are UTF-16: a char outside the BMP is two surrogate code units, each of which
    Encoding.UTF8.GetBytes turns into U+FFFD (EF BF BD)."""
    for ch in s:
        if ord(ch) > 0xFFFF:
            yield "�"
            yield "�"
        else:
            yield ch


class Tokenizer:
    def __init__(self, metadata: dict):
        raw = metadata["tokenizer.ggml.tokens"]
        self.tokens = [t.encode("utf-8") if isinstance(t, str) else bytes(t) for t in raw]
        self.merges = []
        for m in metadata["tokenizer.ggml.merges"]:
            sp = m.index(" ")
            self.merges.append((m[:sp].encode("utf-8"), m[sp + 1:].encode("utf-8")))
        # first occurrence keeps the lowest rank, as the reference's linear scan finds it first
        self._rank = {}
        for k, pair in enumerate(self.merges):
            self._rank.setdefault(pair, k)
        self.BosTokenId = int(metadata["tokenizer.ggml.bos_token_id"])
        self.EosTokenId = int(metadata["tokenizer.ggml.eos_token_id"])
        self.byteSequenceToId = {}
        for i, t in enumerate(self.tokens):
            self.byteSequenceToId[t] = i  # later duplicates overwrite, as Dictionary[key] = value does
        self.idToByteSequence = dict(enumerate(self.tokens))

    # -- Tokenizer.cs:242-267
    @staticmethod
    def ToInitialBpeUnits(text: str) -> list[bytes]:
        out = []
        for ch in _utf16_units(text):
            if ch == " ":
                out.append(b"\xc4\xa0")
            elif ch == "\n":
                out.append(b"\xc4\x8a")
            else:
                out.append(ch.encode("utf-8", "replace") if not ("\ud800" <= ch <= "\udfff") else b"\xef\xbf\xbd")
        return out

    def _bpe(self, parts: list[bytes]) -> list[bytes]:
        parts = list(parts)
        while True:
            best_k, best_j = -1, -1
            for j in range(len(parts) - 1):
                k = self._rank.get((parts[j], parts[j + 1]))
                if k is not None and (best_k == -1 or k < best_k):
                    best_k, best_j = k, j
            if best_k == -1:
                return parts
            parts[best_j:best_j + 2] = [parts[best_j] + parts[best_j + 1]]

    @staticmethod
    def Template(prompt: str, addBos: bool) -> str:
        if addBos:
            t = ("<|begin_of_text|><|start_header_id|>system<|end_header_id|>\n\n"
                 f"{SYSTEM_PROMPT}<|eot_id|><|start_header_id|>user<|end_header_id|>\n\n"
                 f"{prompt}<|eot_id|><|start_header_id|>assistant<|end_header_id|>\n\n")
        else:
            t = ("\n\n<|start_header_id|>user<|end_header_id|>\n\n"
                 f"{prompt}<|eot_id|><|start_header_id|>assistant<|end_header_id|>\n\n")
        return t.replace("\r", "")

    def Tokenize(self, prompt: str, addBos: bool = True, addEos: bool = False) -> list[int]:
        template = self.Template(prompt, addBos)
        ids: list[int] = []
        matches = list(_SPECIAL.finditer(template))
        # text BEFORE the first special token is dropped by the reference's loop (Tokenizer.cs:99-107):
        # only (special token, text until the next special token) pairs are tokenized
        for i, m in enumerate(matches):
            nxt = matches[i + 1].start() if i + 1 < len(matches) else len(template)
            special = m.group(0).encode("utf-8")
            if special not in self.byteSequenceToId:
                raise KeyError(f"special token {m.group(0)!r} not in vocabulary")
            ids.append(self.byteSequenceToId[special])
            text = template[m.end():nxt]
            for piece in _PRETOK.finditer(text):
                for part in self._bpe(self.ToInitialBpeUnits(piece.group(0))):
                    tid = self.byteSequenceToId.get(part)
                    if tid is None:
                        raise KeyError(f"Token not found: {part.decode('utf-8', 'replace')}")
                    ids.append(tid)
        return ids  # addEos is accepted and ignored, as in the reference's live code path

    def Detokenize(self, tokenIds) -> str:
        b = bytearray()
        for i in tokenIds:
            if int(i) not in self.idToByteSequence:
                raise KeyError(f"Token ID {i} not found in vocabulary.")
            b += self.idToByteSequence[int(i)]
        return b.decode("utf-8", "replace").replace("Ġ", " ").replace("Ċ", "\n")

    def BuildChatPrompt(self, messages: list[str]) -> list[int]:
        return self.Tokenize("".join(messages) + "<|start_header_id|>assistant<|end_header_id|>\n\n", addBos=True)
