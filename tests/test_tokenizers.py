"""Caption front ends (frido_amd/tokenizers.py) against the installed third-party implementations on SYNTHETIC vocabularies.

The reference tokenizes with HF `BertTokenizerFast("bert-base-uncased")` (frido/modules/encoders/modules.py:59-73) and with
`clip.tokenize` (modules.py:208).  Their vocabulary FILES are downloads (unreachable here), but the ALGORITHMS are installed with
`transformers` / `tokenizers`, so the restatements are pinned on vocabularies the test writes itself: WordPiece bit-exact against
the Rust BertWordPieceTokenizer and BertTokenizerFast, the byte-level BPE against CLIPTokenizer.  CPU only.
"""
import gzip
import json

import pytest
import torch

from frido_amd.tokenizers import WordPieceTokenizer, ClipBPETokenizer, bytes_to_unicode

AZ = "abcdefghijklmnopqrstuvwxyz0123456789"
BERT_VOCAB = (["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list(AZ) + ["##" + c for c in AZ] +
              ["the", "cat", "dog", "sitting", "sit", "##ting", "on", "mat", ",", ".", "!", "'", "-", "$", "caf", "人",
               "two", "people", "##le", "peop", "stand", "##ing", "next", "to", "tree", "##es", "un", "##believ", "##able", "?", "(", ")"])
CAPTIONS = [
    "The cats, sitting on a Café-mat! 人x",
    "two people standing next to trees.",
    "  UNBELIEVABLE\tdogs?  (the  $5 mat)  ",
    "a zebra with é́ accents and \x00control\x07 chars �",
    "~tilde `back^tick` | pipe",
    "",
    "a " * 100,
    "x" * 120 + " cat",
    "[MASK] the [SEP] cat",
    "日本語 の 人人",
]


@pytest.fixture(scope="module")
def bert_vocab(tmp_path_factory):
    p = tmp_path_factory.mktemp("tk") / "vocab.txt"
    p.write_text("\n".join(BERT_VOCAB) + "\n", encoding="utf-8")
    return str(p)


def test_wordpiece_matches_rust_bert_wordpiece(bert_vocab):
    from tokenizers import BertWordPieceTokenizer
    ref = BertWordPieceTokenizer(bert_vocab, lowercase=True)
    mine = WordPieceTokenizer(bert_vocab)
    assert mine.vocab_size == len(BERT_VOCAB) == len(set(BERT_VOCAB))
    for c in CAPTIONS:
        assert [mine.cls] + mine.encode(c) + [mine.sep] == ref.encode(c).ids, c


@pytest.mark.parametrize("max_length", [8, 16, 77])
def test_wordpiece_call_matches_the_reference_call_of_bert_tokenizer_fast(bert_vocab, max_length):
    """The exact call of encoders/modules.py:69-70 (truncation, padding='max_length', return_tensors='pt') -> input_ids."""
    from transformers import BertTokenizerFast
    hf = BertTokenizerFast(vocab=bert_vocab, do_lower_case=True)
    if hf.vocab_size != len(BERT_VOCAB):
        pytest.skip("this transformers version did not load the synthetic vocabulary")
    mine = WordPieceTokenizer(bert_vocab)
    want = hf(CAPTIONS, truncation=True, max_length=max_length, return_length=True, return_overflowing_tokens=False,
              padding="max_length", return_tensors="pt")["input_ids"]
    got = mine(CAPTIONS, max_length=max_length)
    assert got.dtype == torch.long and torch.equal(got, want)
    assert torch.equal(mine(CAPTIONS[0], max_length=max_length), want[:1])         # a single string is a batch of one


def test_bert_embedder_tokenizes_strings_from_a_local_vocab(bert_vocab, monkeypatch):
    from frido_amd.models import BERTEmbedder
    m = BERTEmbedder(n_embed=32, n_layer=1, vocab_size=len(BERT_VOCAB), max_seq_len=12, vocab_file=bert_vocab)
    ids = m._tokenize(["the cat", "a dog sitting on the mat."])
    assert ids.shape == (2, 12) and ids[0, 0] == 2 and ids[0, 3] == 3 and int(ids[0, 4:].abs().sum()) == 0
    m2 = BERTEmbedder(n_embed=32, n_layer=1, vocab_size=len(BERT_VOCAB), max_seq_len=12)
    monkeypatch.setenv("FRIDO_BERT_VOCAB", bert_vocab)
    assert torch.equal(m2._tokenize(["the cat"]), ids[:1])
    monkeypatch.delenv("FRIDO_BERT_VOCAB")
    m3 = BERTEmbedder(n_embed=32, n_layer=1, vocab_size=len(BERT_VOCAB), max_seq_len=12)
    with pytest.raises(NotImplementedError, match="FRIDO_BERT_VOCAB"):
        m3._tokenize(["the cat"])


# ---------------------------------------------------------------------------------------------------------------------
MERGES = ["t h", "th e</w>", "c a", "ca t</w>", "a t</w>", "s i", "si t", "sit t", "i n", "in g</w>", "sitt ing</w>", "o n</w>", "d o",
          "do g</w>", "m at</w>", "1 2", "' s</w>", "t r", "tr e", "tre e</w>", "e s</w>", "! !", "!! !</w>", "Ã ©</w>"]
CLIP_CAPTIONS = ["the cat sitting on the mat", "The Cat's dog!!!", "a   tree\n and 12 trees", "trees, dogs & cats", "café",
                 "", "x" * 20]


@pytest.fixture(scope="module")
def clip_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("clip")
    merges = d / "merges.txt"
    merges.write_text("#version: 0.2\n" + "\n".join(MERGES) + "\n", encoding="utf-8")
    gz = d / "bpe_simple_vocab_16e6.txt.gz"          # the OpenAI file's form: a header line, then one merge per line, gzipped
    with gzip.open(gz, "wt", encoding="utf-8") as f:
        f.write('"#version: 0.2\n' + "\n".join(MERGES) + "\n")
    b = list(bytes_to_unicode().values())
    vocab = b + [v + "</w>" for v in b] + ["".join(m.split()) for m in MERGES] + ["<|startoftext|>", "<|endoftext|>"]
    vj = d / "vocab.json"
    vj.write_text(json.dumps({t: i for i, t in enumerate(vocab)}), encoding="utf-8")
    return str(merges), str(gz), str(vj)


def test_clip_bpe_matches_hf_clip_tokenizer(clip_files):
    merges, gz, vj = clip_files
    from transformers import CLIPTokenizer
    try:
        hf = CLIPTokenizer(vocab=vj, merges=merges)
    except Exception as e:      # older spelling
        pytest.skip(f"CLIPTokenizer could not be built from local files: {e}")
    for path in (merges, gz):
        mine = ClipBPETokenizer(path)
        assert len(mine.encoder) == 512 + len(MERGES) + 2
        for c in CLIP_CAPTIONS:
            want = hf(c)["input_ids"]
            got = [mine.sot] + mine.encode(c) + [mine.eot]
            assert got == want, (c, got, want)


def test_clip_tokenize_layout_and_errors(clip_files):
    """clip.tokenize: [B, 77] int64, <|startoftext|> ids <|endoftext|> then zeros; RuntimeError on an over-long caption."""
    mine = ClipBPETokenizer(clip_files[1])
    t = mine(["the cat", "a dog"])
    assert t.shape == (2, 77) and t.dtype == torch.long
    assert t[0, 0] == mine.sot and t[0, 3] == mine.eot and int(t[0, 4:].sum()) == 0
    assert mine("the cat").shape == (1, 77)
    with pytest.raises(RuntimeError, match="too long"):
        mine("z " * 100)
    tr = mine("z " * 100, truncate=True)
    assert tr[0, -1] == mine.eot and tr[0, 0] == mine.sot
    assert mine.encode("trees, dogs &amp;amp; cats") == mine.encode("trees, dogs & cats")      # html.unescape twice (OpenAI's basic_clean; HF's does not)
    # the merge-count cap of the OpenAI file (first 49152 - 256 - 2 merges) is honoured
    assert len(ClipBPETokenizer(clip_files[0], n_merges=3).ranks) == 3


def test_clip_text_embedder_tokenizes_strings_from_a_local_merge_table(clip_files, monkeypatch):
    from frido_amd.models import FrozenCLIPTextEmbedder
    arch = (16, 77, 512 + len(MERGES) + 2, 32, 2, 1)
    m = FrozenCLIPTextEmbedder(arch=arch, bpe_path=clip_files[1])
    ids = m._tokens(["the cat"])
    assert ids.shape == (1, 77) and int(ids.argmax(-1)) == 3        # the <|endoftext|> id is the largest: encode_text's pooling index
    monkeypatch.setenv("FRIDO_CLIP_BPE", clip_files[0])
    assert torch.equal(FrozenCLIPTextEmbedder(arch=arch)._tokens(["the cat"]), ids)
    monkeypatch.delenv("FRIDO_CLIP_BPE")
    try:
        import clip  # noqa: F401
    except ImportError:
        with pytest.raises(NotImplementedError, match="FRIDO_CLIP_BPE"):
            FrozenCLIPTextEmbedder(arch=arch)._tokens(["the cat"])


# ---------------------------------------------------------------------------------------------------------------------
# random captions (hypothesis): the restatements against the installed implementations beyond the hand-written cases
def test_wordpiece_matches_rust_on_random_unicode_captions(bert_vocab):
    from hypothesis import given, settings, strategies as st
    from tokenizers import BertWordPieceTokenizer
    ref = BertWordPieceTokenizer(bert_vocab, lowercase=True)
    mine = WordPieceTokenizer(bert_vocab)
    alphabet = st.sampled_from(list("abcdefgxyzTHECATS0159 \t\n,.!?'()-$~^`|éÉüñçÅ人日本́  \x07​�_[]#"))
    words = st.sampled_from(["the", "cat", "dog", "sitting", "unbelievable", "standing", "people", "trees", "[MASK]", "[SEP]", "##ing", "café"])

    @settings(max_examples=300, deadline=None, database=None)
    @given(st.lists(st.one_of(st.text(alphabet, max_size=12), words), max_size=12).map(" ".join))
    def check(caption):
        assert [mine.cls] + mine.encode(caption) + [mine.sep] == ref.encode(caption).ids, repr(caption)

    check()


def test_clip_bpe_matches_hf_on_random_ascii_captions(clip_files):
    from hypothesis import given, settings, strategies as st
    from transformers import CLIPTokenizer
    merges, gz, vj = clip_files
    hf = CLIPTokenizer(vocab=vj, merges=merges)
    mine = ClipBPETokenizer(gz)
    alphabet = st.sampled_from(list("abcdeghimnorstTHECATS0129 ,.!?'-"))
    words = st.sampled_from(["the", "cat", "cat's", "dog", "sitting", "on", "mat", "tree", "trees", "12", "!!!", "it's", "we're", "i'd"])

    @settings(max_examples=200, deadline=None, database=None)
    @given(st.lists(st.one_of(st.text(alphabet, max_size=10), words), max_size=10).map(" ".join))
    def check(caption):
        assert [mine.sot] + mine.encode(caption) + [mine.eot] == hf(caption)["input_ids"], repr(caption)

    check()


def test_wordpiece_special_tokens_anywhere_like_bert_tokenizer_fast(bert_vocab, tmp_path):
    """(r06, advisor) HF's added-token matching pulls a special token out of the text wherever it stands -- "a[SEP]b", "cats [SEP]." -- not
    only as a whitespace-separated word; a vocabulary WITHOUT [MASK] must treat the string as ordinary text instead of raising KeyError."""
    from transformers import BertTokenizerFast
    hf = BertTokenizerFast(vocab=bert_vocab, do_lower_case=True)
    if hf.vocab_size != len(BERT_VOCAB):
        pytest.skip("this transformers version did not load the synthetic vocabulary")
    mine = WordPieceTokenizer(bert_vocab)
    for c in ("a[SEP]b", "cats [SEP].", "[CLS]a cat[PAD][PAD] on", "a [UNK]cat", "x[SEP][SEP]y [SEP]"):
        want = hf(c, add_special_tokens=True)["input_ids"]
        assert [mine.cls] + mine.encode(c) + [mine.sep] == want, c
    nomask = tmp_path / "vocab_nomask.txt"
    nomask.write_text("\n".join(t for t in BERT_VOCAB if t != "[MASK]") + "\n", encoding="utf-8")
    tk = WordPieceTokenizer(str(nomask))
    assert "[MASK]" not in tk.special and isinstance(tk.encode("a [MASK] cat"), list)       # ordinary text: "[", "mask", "]" pieces or [UNK]s, no KeyError


def test_clip_merges_file_without_a_header_keeps_its_first_merge(tmp_path):
    """(r06, advisor) only a real header line is dropped ("#version: ..." / the OpenAI file's first line): a header-less merges file whose first
    merge starts with '#' or '"' keeps that merge."""
    from frido_amd.tokenizers import ClipBPETokenizer
    body = ['# a', 'a b</w>', '" x</w>']
    for header, n in ((None, 3), ("#version: 0.2", 3), ('"bpe_simple_vocab_16e6.txt#version: 0.2', 3)):
        p = tmp_path / f"m_{n}_{bool(header)}_{len(header or '')}.txt"
        p.write_text("\n".join(([header] if header else []) + body) + "\n", encoding="utf-8")
        tk = ClipBPETokenizer(str(p), n_merges=None)
        assert len(tk.ranks) == n and ("#", "a") in tk.ranks, header
