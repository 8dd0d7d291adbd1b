"""CLIP byte-pair tokenizer for the prompt side: same surface as the reference's
utils/tokenizer.py:64-163 (`SimpleTokenizer(bpe_path)`, `.encode`, `.decode`, `tokenizer(texts, context_length)`),
written from the published algorithm (Radford et al. 2021, CLIP; the GPT-2 byte-level BPE of Sennrich et al. 2016).

The merge table is the public CLIP vocabulary file `bpe_simple_vocab_16e6.txt.gz` (1.3 MB, not shipped here).  It is
looked up, in this order, at: the `bpe_path` argument, $PPT_BPE_VOCAB, ./utils/bpe_simple_vocab_16e6.txt.gz (the
reference's layout, relative to the working directory), ppt_amd/data/.  The class lists of the reference's datasets do
not need it: their token ids are a committed fixture (ppt_amd/data/classnames.json, models/ULIP_models.py here).

Algorithm: clean the text (html entities, whitespace runs, lower case), split it with the CLIP pattern (special tokens,
English contractions, letter runs, single digits, punctuation runs), map each piece's UTF-8 bytes to the 256 printable
stand-in characters, then repeatedly fuse the adjacent symbol pair with the lowest merge rank until none is ranked.
The last symbol of every piece carries the end-of-word mark `</w>`.  Vocabulary ids: 256 byte symbols, the same 256
with `</w>`, the first 48 894 merges in file order, `<|startoftext|>` = 49406, `<|endoftext|>` = 49407.
"""
import gzip
import html
import os

import regex

VOCAB_FILE = "bpe_simple_vocab_16e6.txt.gz"
N_MERGES = 49152 - 256 - 2
SOT, EOT = "<|startoftext|>", "<|endoftext|>"
END = "</w>"
_SPLIT = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                       regex.IGNORECASE)


def find_vocab(bpe_path=None):
    """Path of the merge file, or None."""
    here = os.path.dirname(os.path.abspath(__file__))
    for cand in (bpe_path, os.environ.get("PPT_BPE_VOCAB"), os.path.join("utils", VOCAB_FILE),
                 os.path.join(here, "data", VOCAB_FILE)):
        if cand and os.path.exists(cand):
            return cand
    return None


def byte_symbols():
    """The 256 stand-in characters: printable Latin-1 bytes keep their own code point, the other 68 bytes take
    U+0100, U+0101, ... in byte order (so no symbol is whitespace or a control character)."""
    keep = set(range(0x21, 0x7F)) | set(range(0xA1, 0xAD)) | set(range(0xAE, 0x100))
    table, spare = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + spare)
            spare += 1
    return table


class SimpleTokenizer:
    def __init__(self, bpe_path=None):
        path = find_vocab(bpe_path)
        if path is None:
            raise FileNotFoundError(
                f"CLIP merge table {VOCAB_FILE} not found (pass bpe_path=, set PPT_BPE_VOCAB, or place it under ./utils/); "
                "only class names outside ppt_amd/data/classnames.json need it")
        with gzip.open(path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        merges = [tuple(l.split()) for l in lines[1:1 + N_MERGES]]            # line 0 is the file's version header
        self.byte_sym = byte_symbols()
        self.sym_byte = {c: b for b, c in self.byte_sym.items()}
        # id order of the vocabulary: byte symbols sorted the way the table above lists them (kept bytes first, in byte
        # order, then the 68 re-mapped ones), the same with the end-of-word mark, the merges, the two specials
        kept = [b for b in range(256) if ord(self.byte_sym[b]) < 256]
        moved = [b for b in range(256) if ord(self.byte_sym[b]) >= 256]
        base = [self.byte_sym[b] for b in kept + moved]
        vocab = base + [s + END for s in base] + [a + b for a, b in merges] + [SOT, EOT]
        self.encoder = {s: i for i, s in enumerate(vocab)}
        self.decoder = {i: s for s, i in self.encoder.items()}
        self.rank = {pair: i for i, pair in enumerate(merges)}
        self._memo = {SOT: [SOT], EOT: [EOT]}

    # ---- BPE on one piece ------------------------------------------------------------------------
    def _merge(self, piece):
        hit = self._memo.get(piece)
        if hit is not None:
            return hit
        syms = list(piece[:-1]) + [piece[-1] + END]
        while len(syms) > 1:
            best, where = None, -1
            for i in range(len(syms) - 1):                 # lowest-ranked adjacent pair, first occurrence
                r = self.rank.get((syms[i], syms[i + 1]))
                if r is not None and (best is None or r < best):
                    best, where = r, i
            if best is None:
                break
            a, b = syms[where], syms[where + 1]
            fused, i = [], 0
            while i < len(syms):                           # every non-overlapping occurrence, left to right
                if i + 1 < len(syms) and syms[i] == a and syms[i + 1] == b:
                    fused.append(a + b)
                    i += 2
                else:
                    fused.append(syms[i])
                    i += 1
            syms = fused
        self._memo[piece] = syms
        return syms

    # ---- reference surface -------------------------------------------------------------------------
    @staticmethod
    def clean(text):
        try:
            import ftfy                                    # the reference repairs mojibake first; optional here
            text = ftfy.fix_text(text)
        except ImportError:
            pass
        text = html.unescape(html.unescape(text))
        return regex.sub(r"\s+", " ", text.strip()).strip().lower()

    def encode(self, text):
        ids = []
        for piece in _SPLIT.findall(self.clean(text)):
            mapped = "".join(self.byte_sym[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[s] for s in self._merge(mapped))
        return ids

    def decode(self, tokens):
        text = "".join(self.decoder[int(t)] for t in tokens)
        return bytearray(self.sym_byte[c] for c in text).decode("utf-8", errors="replace").replace(END, " ")

    def __call__(self, texts, context_length=77):
        import torch
        rows = []
        for t in ([texts] if isinstance(texts, str) else texts):
            ids = [self.encoder[SOT]] + self.encode(t) + [self.encoder[EOT]]
            ids = ids[:context_length]                     # (the reference truncates without re-appending EOT)
            rows.append(ids + [0] * (context_length - len(ids)))
        out = torch.tensor(rows, dtype=torch.long)
        return out[0] if len(rows) == 1 else out            # utils/tokenizer.py:161-163: one row -> [context_length]
