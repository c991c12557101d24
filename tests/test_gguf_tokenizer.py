"""Loader-side rows of the scope table on CPU: GGUF reader/writer and the
reference-faithful tokenizer.  The reference holds no fixtures for either; known answers below are
derived by hand from the reference's code (cited)."""
import numpy as np
import pytest

from nfai_amd import gguf, synth
from nfai_amd.llama_model import QuantTensor
from nfai_amd.tokenizer import Tokenizer


def test_gguf_roundtrip_all_value_types_and_alignment(tmp_path):
    import oracle as orc
    rng = np.random.default_rng(1)
    w = gguf.GGUFWriter(alignment=64)
    w.add("general.architecture", "llama")
    w.add("llama.block_count", 2)
    w.add("llama.attention.layer_norm_rms_epsilon", 1e-5)
    w.add("some.bool", True)
    w.add("some.i64", -5, gguf.T_I64)
    w.add("some.u64", 2 ** 40, gguf.T_U64)
    w.add("some.f64", 0.25, gguf.T_F64)
    w.add("some.u8", 7, gguf.T_U8)
    w.add("tokenizer.ggml.tokens", ["a", "b", "Ġc", "<|eot_id|>"])
    w.add("some.ints", [1, 2, 3])
    w.add("some.floats", [0.5, 1.5])
    a32 = rng.standard_normal((3, 8)).astype(np.float32)
    a16 = rng.standard_normal((5, 16)).astype(np.float16)
    v = rng.standard_normal(7).astype(np.float32)
    q4 = orc.quantize_q4k(rng.standard_normal((2, 512)).astype(np.float32))
    q6 = orc.quantize_q6k(rng.standard_normal((3, 256)).astype(np.float32))
    w.add_tensor("a32", a32)
    w.add_tensor("a16", a16)
    w.add_tensor("v", v)
    w.add_tensor("q4", QuantTensor(q4, 12, (2, 512)))
    w.add_tensor("q6", QuantTensor(q6, 14, (3, 256)))
    path = str(tmp_path / "t.gguf")
    w.write(path)
    p = gguf.Parser()
    md, t = p.Read(path)
    assert p.version == 3 and p.alignment == 64
    assert md["general.architecture"] == "llama" and md["llama.block_count"] == 2 and md["some.bool"] is True
    assert md["some.i64"] == -5 and md["some.u64"] == 2 ** 40 and md["some.f64"] == 0.25 and md["some.u8"] == 7
    assert md["tokenizer.ggml.tokens"] == ["a", "b", "Ġc", "<|eot_id|>"] and md["some.ints"] == [1, 2, 3]
    assert abs(md["llama.attention.layer_norm_rms_epsilon"] - 1e-5) < 1e-12
    np.testing.assert_array_equal(t["a32"], a32)
    np.testing.assert_array_equal(t["a16"], a16)
    np.testing.assert_array_equal(t["v"], v)
    assert t["q4"].shape == (2, 512) and t["q4"].ggml_type == 12 and t["q6"].shape == (3, 256)
    np.testing.assert_array_equal(t["q4"].data, q4)
    np.testing.assert_array_equal(t["q6"].data, q6)
    for ti in p.GetTensorInfo():
        assert ti.data_offset % 64 == 0  # Parser.cs:47-59, :125-128
    assert [ti.shape for ti in p.GetTensorInfo()][:2] == [(8, 3), (16, 5)]  # GGUF order: ne0 first
    assert p.GetTensorNames() == ["a32", "a16", "v", "q4", "q6"]


def test_gguf_rejects_bad_magic_and_unsupported_types(tmp_path):
    bad = tmp_path / "bad.gguf"
    bad.write_bytes(b"GGML" + b"\0" * 64)
    with pytest.raises(ValueError, match="Invalid GGUF"):
        gguf.Parser().Read(str(bad))
    w = gguf.GGUFWriter()
    w.add("general.architecture", "llama")
    w.add_tensor("q8", np.zeros(34, np.uint8), ggml_type=8, shape=(1, 32))  # Q8_0: named by Parser.cs:270, no kernel
    path = str(tmp_path / "q8.gguf")
    w.write(path)
    with pytest.raises(ValueError, match="Unsupported data type"):
        gguf.Parser().Read(path)


def test_factory_selection_order(tmp_path):
    """First factory whose TryCreate returns true wins (Parser.cs:36-42)."""
    from nfai_amd.llama_model import ModelOptions
    dims = synth.TINY
    path = str(tmp_path / "m.gguf")
    gguf.write_model(path, synth.make_metadata(dims), synth.make_weights(dims, seed=3))

    class No:
        def TryCreate(self, md, t, opts, **kw):
            return False, None

    class Yes:
        def __init__(self, tag):
            self.tag = tag

        def TryCreate(self, md, t, opts, **kw):
            assert md["general.architecture"] == "llama" and t["blk.1.ffn_up.weight"].shape == (dims.F, dims.E)
            return True, self.tag

    assert gguf.Parser([No(), Yes("second"), Yes("third")]).Parse(ModelOptions(GGUFPath=path)) == "second"
    with pytest.raises(RuntimeError, match="No suitable model factory"):
        gguf.Parser([No()]).Parse(ModelOptions(GGUFPath=path))
    with pytest.raises(FileNotFoundError):
        gguf.Parser([No()]).Parse(ModelOptions(GGUFPath=path + ".missing"))


# ---- tokenizer --------------------------------------------------------------------------------
def _vocab():
    specials = ["<|begin_of_text|>", "<|start_header_id|>", "<|end_header_id|>", "<|eot_id|>"]
    chars = list("abcdefghijklmnopqrstuvwxyzYHW.,!?'0123456789") + ["Ġ", "Ċ", "é"]
    merges = ["h e", "l l", "he ll", "hell o", "Ġ w", "o r", "Ġw or", "Ċ Ċ", "Ġ a", "1 2", "12 3"]
    merged = [m.replace(" ", "") for m in merges]
    toks = specials + chars + merged
    md = {"tokenizer.ggml.tokens": toks, "tokenizer.ggml.merges": merges,
          "tokenizer.ggml.bos_token_id": 0, "tokenizer.ggml.eos_token_id": 3}
    return Tokenizer(md), {t: i for i, t in enumerate(toks)}


def test_bpe_units_and_merge_order():
    tok, ids = _vocab()
    # ' ' -> 'Ġ' (C4 A0), '\n' -> 'Ċ' (C4 8A); everything else its own UTF-8 bytes (Tokenizer.cs:242-267)
    assert Tokenizer.ToInitialBpeUnits("a b\n") == [b"a", b"\xc4\xa0", b"b", b"\xc4\x8a"]
    assert Tokenizer.ToInitialBpeUnits("é") == ["é".encode()]
    assert Tokenizer.ToInitialBpeUnits("\U0001F600") == [b"\xef\xbf\xbd"] * 2  # surrogate pair -> two U+FFFD
    # lowest-rank merge first: h e -> he, l l -> ll, he ll -> hell, hell o -> hello
    assert tok._bpe(tok.ToInitialBpeUnits("hello")) == [b"hello"]
    assert tok._bpe(tok.ToInitialBpeUnits(" world")) == ["Ġwor".encode(), b"l", b"d"]
    assert tok._bpe(tok.ToInitialBpeUnits("lll")) == [b"ll", b"l"]  # leftmost pair wins on equal rank


def test_template_and_token_stream():
    tok, ids = _vocab()
    out = tok.Tokenize("hello world", addBos=True)
    s = lambda t: ids[t]
    head = [s("<|begin_of_text|>"), s("<|start_header_id|>")] + [s(c) for c in "system"] + [s("<|end_header_id|>"), s("ĊĊ")]
    assert out[:len(head)] == head
    # user turn: "\n\nhello world" -> ĊĊ | hello | Ġwor l d
    user = [s("<|start_header_id|>")] + [s(c) for c in "user"] + [s("<|end_header_id|>"), s("ĊĊ"), s("hello"), s("Ġwor"), s("l"), s("d"), s("<|eot_id|>")]
    assert any(out[i:i + len(user)] == user for i in range(len(out)))
    tail = [s("<|start_header_id|>")] + [s(c) for c in "assistant"] + [s("<|end_header_id|>"), s("ĊĊ")]
    assert out[-len(tail):] == tail
    assert s("<|begin_of_text|>") not in tok.Tokenize("hello", addBos=False)  # continuation template (Tokenizer.cs:84-90)
    # \r is stripped from the whole template (Tokenizer.cs:80)
    assert tok.Tokenize("hello\r", addBos=True) == tok.Tokenize("hello", addBos=True)
    # the continuation template starts with text before the first special token, which the reference drops
    cont = tok.Tokenize("hello", addBos=False)
    assert cont[0] == s("<|start_header_id|>")
    # digits split in groups of <= 3 by the pre-tokenizer, then merged: 1234 -> "123" + "4"
    assert [ids["123"], ids["4"]] == [t for t in tok.Tokenize("1234", True) if t in (ids["123"], ids["4"])]
    with pytest.raises(KeyError, match="Token not found"):
        tok.Tokenize("Z")  # 'Z' is not in the vocabulary (Tokenizer.cs:174-177)


def test_detokenize():
    tok, ids = _vocab()
    assert tok.Detokenize([ids["hello"], ids["Ġwor"], ids["l"], ids["d"], ids["ĊĊ"]]) == "hello world\n\n"
    with pytest.raises(KeyError):
        tok.Detokenize([10 ** 6])
    out = tok.Tokenize("hello, world!", True)
    text = tok.Detokenize(out)
    assert "hello, world!" in text and text.startswith("<|begin_of_text|>")


def test_bpe_and_pretokenizer_against_the_tokenizers_library():
    """The merge loop (lowest rank first, leftmost on ties; Tokenizer.cs:130-166) and the Llama-3 split pattern (Tokenizer.cs:93-111)
    against an implementation that shares no code with this repository or the reference: the `tokenizers` library's byte-level BPE with
    the same split pattern, on a vocabulary and merge list trained by that library from a small corpus (nothing is downloaded).  Printable
    ASCII, spaces and newlines only: for those the reference's initial units (' ' -> 'Ġ', '\\n' -> 'Ċ', every other character itself;
    Tokenizer.cs:242-267) coincide with the GPT-2 byte map — a tab or a non-ASCII character does not (kept as the reference has it, see
    the module docstring), so those are outside this cross-check."""
    import json
    tk = pytest.importorskip("tokenizers")
    from tokenizers import Regex, models, pre_tokenizers, trainers
    from nfai_amd.tokenizer import Tokenizer, _PRETOK
    corpus = ["The quick brown fox jumps over the lazy dog. " * 3, "def main():\n    print('hello, world')\n    return 0\n",
              "It's 2024 and we've got 1234567 reasons; they're all fine!  Aren't they?\n\nYes -- they'll do.",
              "tokenization of repeated tokens tokens tokens", "  leading spaces and   runs   of   spaces \n \n trailing "]
    hf = tk.Tokenizer(models.BPE())
    hf.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(Regex(_PRETOK.pattern), behavior="isolated"),
                                                pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    hf.train_from_iterator(corpus * 4, trainers.BpeTrainer(vocab_size=400, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(),
                                                           special_tokens=[], show_progress=False))
    model = json.loads(hf.to_str())["model"]
    toks = [None] * len(model["vocab"])
    for t, i in model["vocab"].items():
        toks[i] = t
    merges = [m if isinstance(m, str) else " ".join(m) for m in model["merges"]]
    assert len(merges) > 100
    mine = Tokenizer({"tokenizer.ggml.tokens": toks, "tokenizer.ggml.merges": merges, "tokenizer.ggml.bos_token_id": 0,
                      "tokenizer.ggml.eos_token_id": 0})
    texts = corpus + ["Hello there, world!  What's   up?\nfine\n\n\nok 12 123 1234 12345", "a", " ", "", "x  y", "don't DON'T 'S 'Ll",
                      "the the the then there other", "trailing newline\n", "\n\n\nleading newlines", "a.b,c;d:e!f?g(h)i[j]k{l}m"]
    for t in texts:
        got = [mine.byteSequenceToId[part] for piece in _PRETOK.finditer(t) for part in mine._bpe(mine.ToInitialBpeUnits(piece.group(0)))]
        assert got == hf.encode(t).ids, t
        assert mine.Detokenize(got) == t
