"""`clip.tokenize` of the reference's call sites (AvatarGen/AppearanceGen/main.py:274,280,286): OpenAI CLIP's byte-level BPE
(third-party package `clip`, clip/simple_tokenizer.py; not vendored).  The algorithm is restated here; the merge table
`bpe_simple_vocab_16e6.txt.gz` ships with that package and is NOT available offline, so the path must be supplied
(argument or $AVC_CLIP_BPE).  Checked against transformers' independent CLIP BPE on a synthetic merge table
(tests/test_clip_text.py)."""
import gzip
import html
import os
from functools import lru_cache
from typing import List, Union

import regex as re
import torch


@lru_cache()
def bytes_to_unicode():
    """printable stand-ins for all 256 byte values (the BPE works on these characters, never on raw control bytes)"""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + \
        list(range(ord("®"), ord("ÿ") + 1))
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


def _clean(text):
    try:
        import ftfy
        text = ftfy.fix_text(text)
    except ImportError:      # ftfy only repairs mojibake; plain prompts are unaffected
        pass
    text = html.unescape(html.unescape(text))
    return re.sub(r"\s+", " ", text.strip()).strip()


class SimpleTokenizer:
    def __init__(self, bpe_path: str = None, n_merges: int = 49152 - 256 - 2):
        bpe_path = bpe_path or os.environ.get("AVC_CLIP_BPE")
        if not bpe_path or not os.path.exists(bpe_path):
            raise FileNotFoundError("CLIP's merge table bpe_simple_vocab_16e6.txt.gz is needed to tokenize prompts; pass its path "
                                    "or set AVC_CLIP_BPE (it ships with the OpenAI `clip` package)")
        opener = gzip.open if bpe_path.endswith(".gz") else open
        with opener(bpe_path, "rt", encoding="utf-8") as fp:
            lines = fp.read().split("\n")
        merges = [tuple(l.split()) for l in lines[1:1 + n_merges] if len(l.split()) == 2]
        self.byte_encoder = bytes_to_unicode()
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        vocab += ["".join(m) for m in merges]
        vocab += ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {t: i for i, t in enumerate(vocab)}
        self.decoder = {i: t for t, i in self.encoder.items()}
        self.ranks = {m: i for i, m in enumerate(merges)}
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.pat = re.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                              re.IGNORECASE)

    def bpe(self, token: str) -> str:
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            cand = [p for p in _pairs(word) if p in self.ranks]
            if not cand:
                break
            a, b = min(cand, key=lambda p: self.ranks[p])
            out, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = tuple(out)
        res = " ".join(word)
        self.cache[token] = res
        return res

    def encode(self, text: str) -> List[int]:
        ids = []
        for tok in re.findall(self.pat, _clean(text).lower()):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(tok).split(" "))
        return ids

    def decode(self, ids) -> str:
        text = "".join(self.decoder[int(i)] for i in ids)
        inv = {v: k for k, v in self.byte_encoder.items()}
        return bytearray(inv[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    @property
    def sot(self):
        return self.encoder["<|startoftext|>"]

    @property
    def eot(self):
        return self.encoder["<|endoftext|>"]


def tokenize(texts: Union[str, List[str]], tokenizer: SimpleTokenizer = None, context_length: int = 77,
             truncate: bool = False) -> torch.Tensor:
    """clip.tokenize: [SOT] + bpe + [EOT], zero padded to context_length -> LongTensor [len(texts), context_length]"""
    if isinstance(texts, str):
        texts = [texts]
    tokenizer = tokenizer or SimpleTokenizer()
    out = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, t in enumerate(texts):
        ids = [tokenizer.sot] + tokenizer.encode(t) + [tokenizer.eot]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError("Input %r is too long for context length %d" % (t, context_length))
            ids = ids[:context_length]
            ids[-1] = tokenizer.eot
        out[i, :len(ids)] = torch.tensor(ids)
    return out
