"""open_clip.tokenize for FrozenOpenCLIPEmbedder (ldm/modules/encoders/modules.py:171: `tokens = open_clip.tokenize(text)`).

open_clip is neither in the reference tree nor in this image, and its vocabulary file (bpe_simple_vocab_16e6.txt.gz, 1.3 MB) cannot be
fetched; what is restated here is the published algorithm of its SimpleTokenizer (the byte-level BPE of OpenAI's CLIP), loading the table from
a folder the user provides:

    tok = ClipBPETokenizer.from_folder("/path/with/bpe_simple_vocab_16e6.txt.gz")      # open_clip's own file (.txt or .txt.gz), or
    tok = ClipBPETokenizer.from_folder("/path/with/vocab.json + merges.txt")           # the Hugging Face CLIP tokenizer files
    FrozenOpenCLIPEmbedder(..., tokenizer=tok)           # or tokenizer="/path": the embedder builds it

Algorithm: text -> ftfy.fix_text (absent here: passed through, like captions.py) -> html.unescape twice -> strip -> whitespace runs to one
space -> lower case -> pieces by the CLIP pattern (the special tokens, the English contractions, letter runs, single digits, runs of other
non-space symbols) -> each piece's UTF-8 bytes mapped to printable code points, the last one suffixed "</w>" -> greedy lowest-rank pair
merges -> vocabulary ids; a row is <start_of_text> ids <end_of_text> zero-padded to context_length (77), over-long rows truncated with
<end_of_text> in the last column.
Pinned offline against transformers.CLIPTokenizer on a vocabulary built on the spot (tests/test_host_cpu.py); against open_clip itself:
parity unpinned."""
import gzip
import html
import json
import os
from functools import lru_cache
from typing import List, Sequence, Union

import torch

try:   # the reference environment has ftfy; this image does not
    import ftfy
    _fix = ftfy.fix_text
except ImportError:  # pragma: no cover
    _fix = lambda t: t


@lru_cache()
def bytes_to_unicode():
    """The 256 byte values as printable code points: the visible Latin-1 ranges map to themselves, the rest to 256, 257, ..."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def _pairs(word):
    return {(a, b) for a, b in zip(word[:-1], word[1:])}


class ClipBPETokenizer:
    SOT_TEXT, EOT_TEXT = "<start_of_text>", "<end_of_text>"

    def __init__(self, merges: Sequence[Sequence[str]], encoder: dict = None, context_length: int = 77):
        import regex
        self.byte_encoder = bytes_to_unicode()
        merges = [tuple(m) for m in merges]
        if encoder is None:   # open_clip's construction: bytes, bytes + </w>, the merges in order, the two special tokens
            vocab = list(self.byte_encoder.values())
            vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges] + [self.SOT_TEXT, self.EOT_TEXT]
            encoder = dict(zip(vocab, range(len(vocab))))
        self.encoder = dict(encoder)
        self.bpe_ranks = dict(zip(merges, range(len(merges))))
        self.cache = {self.SOT_TEXT: self.SOT_TEXT, self.EOT_TEXT: self.EOT_TEXT}
        self.pat = regex.compile(r"""<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", regex.IGNORECASE)
        self.sot, self.eot = self.encoder[self.SOT_TEXT], self.encoder[self.EOT_TEXT]
        self.context_length = context_length

    # ---- loading
    @classmethod
    def from_folder(cls, folder: str, context_length: int = 77):
        for name in ("bpe_simple_vocab_16e6.txt.gz", "bpe_simple_vocab_16e6.txt"):
            path = os.path.join(folder, name)
            if os.path.exists(path):
                opener = gzip.open if name.endswith(".gz") else open
                with opener(path, "rt", encoding="utf-8") as f:
                    lines = f.read().split("\n")
                # first line: a version comment; open_clip keeps exactly 49152 - 256 - 2 merges (a 49408-entry vocabulary)
                merges = [tuple(l.split()) for l in lines[1:49152 - 256 - 2 + 1] if l.strip()]
                return cls(merges, None, context_length)
        vj, mt = os.path.join(folder, "vocab.json"), os.path.join(folder, "merges.txt")
        if os.path.exists(vj) and os.path.exists(mt):
            with open(vj, encoding="utf-8") as f:
                enc = json.load(f)
            with open(mt, encoding="utf-8") as f:
                lines = f.read().split("\n")
            merges = [tuple(l.split()) for l in lines if l.strip() and not l.startswith("#version")]
            # the Hugging Face files call the special tokens <|startoftext|> / <|endoftext|>: same ids (the last two)
            enc = {({"<|startoftext|>": cls.SOT_TEXT, "<|endoftext|>": cls.EOT_TEXT}.get(k, k)): v for k, v in enc.items()}
            return cls(merges, enc, context_length)
        raise FileNotFoundError(f"{folder}: neither bpe_simple_vocab_16e6.txt[.gz] (open_clip) nor vocab.json + merges.txt (Hugging Face CLIP tokenizer)")

    # ---- the algorithm
    def bpe(self, token: str) -> str:
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = _pairs(word)
        if not pairs:
            return token + "</w>"
        while True:
            bigram = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new, i = [], 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                except ValueError:
                    new.extend(word[i:])
                    break
                new.extend(word[i:j])
                i = j
                if word[i] == first and i < len(word) - 1 and word[i + 1] == second:
                    new.append(first + second)
                    i += 2
                else:
                    new.append(word[i])
                    i += 1
            word = tuple(new)
            if len(word) == 1:
                break
            pairs = _pairs(word)
        out = " ".join(word)
        self.cache[token] = out
        return out

    @staticmethod
    def clean(text: str) -> str:
        text = html.unescape(html.unescape(_fix(text))).strip()
        return " ".join(text.split()).strip().lower()

    def encode(self, text: str) -> List[int]:
        ids = []
        for piece in self.pat.findall(self.clean(text)):
            piece = "".join(self.byte_encoder[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(piece).split(" "))
        return ids

    def __call__(self, texts: Union[str, Sequence[str]], context_length: int = None) -> torch.LongTensor:
        if isinstance(texts, str):
            texts = [texts]
        n = context_length or self.context_length
        out = torch.zeros(len(texts), n, dtype=torch.long)
        for i, t in enumerate(texts):
            row = [self.sot] + self.encode(t) + [self.eot]
            if len(row) > n:
                row = row[:n]
                row[-1] = self.eot
            out[i, :len(row)] = torch.tensor(row)
        return out
