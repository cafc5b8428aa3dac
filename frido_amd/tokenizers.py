"""Caption front ends of the two text-conditioned Frido configs, from LOCAL vocabulary files (host code, no GPU work).

The reference turns captions into token ids with two third-party tokenizers whose vocabulary files are downloads:
  * `BERTTokenizer` (frido/modules/encoders/modules.py:59-73): HF `BertTokenizerFast.from_pretrained("bert-base-uncased")`,
    called with truncation=True, max_length, padding="max_length"  ->  [B, max_length] int64;
  * `FrozenCLIPTextEmbedder.forward` (modules.py:208): `clip.tokenize(text)` of the un-vendored OpenAI `clip` package
    (simple_tokenizer.py + bpe_simple_vocab_16e6.txt.gz)  ->  [B, 77] int64, zero padded, RuntimeError when a caption is too long.
Neither package's data is reachable offline, so this module restates the two PUBLISHED algorithms (BERT's BasicTokenizer +
greedy longest-match WordPiece; CLIP's byte-level BPE) over a vocabulary file the user points at:
    BERT:  `vocab.txt` of bert-base-uncased   (BERTEmbedder(vocab_file=...) or $FRIDO_BERT_VOCAB)
    CLIP:  `bpe_simple_vocab_16e6.txt.gz` / a plain `merges.txt`   (FrozenCLIPTextEmbedder(bpe_path=...) or $FRIDO_CLIP_BPE)
tests/test_tokenizers.py pins both against the `transformers` implementations (BertTokenizerFast / CLIPTokenizer, which ARE
installed) on synthetic vocabularies written by the test; the real files have never been seen here ("parity unpinned" on them).
"""
import gzip
import html
import os
import re
import unicodedata
from functools import lru_cache

import torch


# ---------------------------------------------------------------------------------------------------------------------
# BERT: BasicTokenizer (clean, lower-case, strip accents, split on punctuation, isolate CJK) + WordPiece
def _is_whitespace(ch):
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch):
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:      # ASCII non-alphanumerics count, e.g. "$", "^", "`"
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class WordPieceTokenizer:
    """`bert-base-uncased`-style tokenizer over a local vocab.txt (one token per line, id = line number)."""

    def __init__(self, vocab_file, do_lower_case=True, unk="[UNK]", cls="[CLS]", sep="[SEP]", pad="[PAD]", max_chars_per_word=100):
        with open(vocab_file, encoding="utf-8") as f:
            toks = [l.rstrip("\n") for l in f]
        self.vocab = {t: i for i, t in enumerate(toks)}
        self.vocab_size = len(toks)
        self.lower = do_lower_case
        for name in (unk, cls, sep, pad):
            if name not in self.vocab:
                raise ValueError(f"{vocab_file}: special token {name} is missing")
        self.unk, self.cls, self.sep, self.pad = (self.vocab[t] for t in (unk, cls, sep, pad))
        # (r06, advisor) "[MASK]" is special only where the vocabulary holds it (a vocab without it used to raise KeyError on a caption containing the string)
        self.special = {t for t in (unk, cls, sep, pad, "[MASK]") if t in self.vocab}
        # HF's added-token matching extracts a special token ANYWHERE in the text ("a[SEP]b", "cats [SEP]."), not only as a whitespace-separated
        # word: the text is split on the special strings first (longest first), the pieces in between go through the basic tokenizer
        self._special_re = re.compile("(" + "|".join(re.escape(t) for t in sorted(self.special, key=len, reverse=True)) + ")")
        self.max_chars = max_chars_per_word

    def _basic(self, text):
        words = []
        for piece in self._special_re.split(text):
            if piece in self.special:
                words.append(piece)
            elif piece:
                words += self._basic_plain(piece)
        return words

    def _basic_plain(self, text):
        out = []
        for ch in text:                                                     # clean: drop NUL / U+FFFD / controls, whitespace -> " "
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_cjk(cp):
                out.append(f" {ch} ")
            else:
                out.append(" " if _is_whitespace(ch) else ch)
        words = []
        for tok in "".join(out).split():
            if self.lower:
                tok = tok.lower()
                tok = "".join(c for c in unicodedata.normalize("NFD", tok) if unicodedata.category(c) != "Mn")
            cur = []
            for ch in tok:                                                  # every punctuation character is its own word
                if _is_punctuation(ch):
                    if cur:
                        words.append("".join(cur))
                        cur = []
                    words.append(ch)
                else:
                    cur.append(ch)
            if cur:
                words.append("".join(cur))
        return words

    def _wordpiece(self, word):
        if len(word) > self.max_chars:
            return [self.unk]
        ids, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:                                              # greedy longest match first
                sub = word[start:end] if start == 0 else "##" + word[start:end]
                if sub in self.vocab:
                    cur = self.vocab[sub]
                    break
                end -= 1
            if cur is None:
                return [self.unk]
            ids.append(cur)
            start = end
        return ids

    def encode(self, text):
        ids = []
        for w in self._basic(text):
            ids += [self.vocab[w]] if w in self.special else self._wordpiece(w)
        return ids

    def __call__(self, text, max_length=77):
        """[CLS] ids [SEP], truncated to max_length (the [SEP] survives), padded with [PAD]: what the reference's call
        (modules.py:69-70: truncation=True, padding="max_length") returns as `input_ids`."""
        if max_length < 2:
            raise ValueError("max_length must leave room for [CLS] and [SEP]")
        texts = [text] if isinstance(text, str) else list(text)
        out = torch.full((len(texts), max_length), self.pad, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.cls] + self.encode(t)[:max_length - 2] + [self.sep]
            out[i, :len(ids)] = torch.tensor(ids, dtype=torch.long)
        return out


# ---------------------------------------------------------------------------------------------------------------------
# CLIP: byte-level BPE (OpenAI clip/simple_tokenizer.py, published with the CLIP paper's code)
@lru_cache()
def bytes_to_unicode():
    """The 256 byte values mapped to printable code points (printable bytes keep theirs, the rest move to 256 + n)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


class ClipBPETokenizer:
    """`clip.tokenize` over a local merge table: the OpenAI `bpe_simple_vocab_16e6.txt.gz` (its first line is a header and only the
    first 49152 - 256 - 2 merges count) or a plain / gzipped HF-style `merges.txt` (header line optional; n_merges=None: all)."""

    def __init__(self, bpe_path, context_length=77, n_merges=49152 - 256 - 2):
        import regex
        opener = gzip.open if str(bpe_path).endswith(".gz") else open
        with opener(bpe_path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        # (r06, advisor) only a REAL header is dropped -- a first line that says "#version": HF's "#version: 0.2", the OpenAI file's '"bpe_simple_vocab_16e6.txt#version: 0.2';
        # a header-less merges file whose first merge happens to start with '#' or '"' keeps it
        if lines and "#version" in lines[0]:
            lines = lines[1:]
        merges = [tuple(l.split()) for l in lines if len(l.split()) == 2]
        if n_merges is not None:
            merges = merges[:n_merges]
        vocab = list(bytes_to_unicode().values())
        vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {t: i for i, t in enumerate(vocab)}
        self.ranks = {m: i for i, m in enumerate(merges)}
        self.byte_enc = bytes_to_unicode()
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.pat = regex.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                                 regex.IGNORECASE)
        self.context_length = context_length
        self.sot, self.eot = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]

    def _bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            pairs = {(word[i], word[i + 1]) for i in range(len(word) - 1)}
            best = min(pairs, key=lambda p: self.ranks.get(p, float("inf")))
            if best not in self.ranks:
                break
            a, b = best
            new, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    new.append(a + b)
                    i += 2
                else:
                    new.append(word[i])
                    i += 1
            word = tuple(new)
        out = " ".join(word)
        self.cache[token] = out
        return out

    @staticmethod
    def _clean(text):
        try:                                                    # the OpenAI tokenizer runs ftfy.fix_text first; absent here
            import ftfy
            text = ftfy.fix_text(text)
        except ImportError:
            pass
        text = html.unescape(html.unescape(text)).strip()
        return " ".join(text.split()).strip().lower()

    def encode(self, text):
        ids = []
        for tok in self.pat.findall(self._clean(text)):
            tok = "".join(self.byte_enc[b] for b in tok.encode("utf-8"))
            ids += [self.encoder[t] for t in self._bpe(tok).split(" ")]
        return ids

    def __call__(self, texts, context_length=None, truncate=False):
        texts = [texts] if isinstance(texts, str) else list(texts)
        n = context_length or self.context_length
        out = torch.zeros(len(texts), n, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > n:
                if not truncate:
                    raise RuntimeError(f"Input {t} is too long for context length {n}")      # clip.tokenize's own error
                ids = ids[:n]
                ids[-1] = self.eot
            out[i, :len(ids)] = torch.tensor(ids, dtype=torch.long)
        return out


def local_bert_vocab(explicit=None):
    """Path of a local BERT vocab.txt: the explicit argument, else $FRIDO_BERT_VOCAB, else None."""
    p = explicit or os.environ.get("FRIDO_BERT_VOCAB")
    return p if p and os.path.isfile(p) else None


def local_clip_bpe(explicit=None):
    p = explicit or os.environ.get("FRIDO_CLIP_BPE")
    return p if p and os.path.isfile(p) else None
