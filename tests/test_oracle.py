"""The CPU oracle (oracle/nfai_oracle.c) against (a) hand-computed known answers and (b) the
independent fp64 NumPy evaluation (oracle/np_oracle.py).  The reference has no golden vectors for
this path (parity unpinned); these tests pin the oracle to the op definitions."""
import os

import numpy as np
import pytest

import oracle as orc
from oracle import np_oracle as npo
from nfai_amd import synth

RNG = np.random.Generator(np.random.PCG64(7))


def test_half_conversion_exhaustive():
    h = np.arange(65536, dtype=np.uint16)
    ref = h.view(np.float16).astype(np.float32)
    got = orc.widen_f16(h.view(np.float16))
    ok = (got == ref) | (np.isnan(got) & np.isnan(ref))
    assert ok.all()
    # narrowing: round-to-nearest-even agrees with NumPy on a dense sample incl. subnormals
    f = np.concatenate([RNG.standard_normal(200000).astype(np.float32) * 10.0 ** RNG.integers(-9, 5, 200000),
                        ref[np.isfinite(ref)]]).astype(np.float32)
    with np.errstate(over="ignore"):
        want = f.astype(np.float16)
    assert (orc.narrow_f16(f).view(np.uint16) == want.view(np.uint16)).all()


def test_known_answers():
    # RMSNorm: x = [3,4], g = [1,2], eps = 0 -> rms = sqrt(12.5)
    y = orc.rmsnorm([3.0, 4.0], [1.0, 2.0], 0.0)
    np.testing.assert_allclose(y, [3 / np.sqrt(12.5), 8 / np.sqrt(12.5)], rtol=1e-6)
    # GEMV: rows are W[j, :]
    W = np.array([[1, 2, 3], [4, 5, 6]], np.float32)
    assert orc.gemv(W, [1, 1, 1]).tolist() == [6.0, 15.0]
    # RoPE at pos 0 is the identity; at pos p pair 0 rotates by p radians (freq[0] = 1)
    x = np.arange(8, dtype=np.float32)
    f = orc.rope_freqs(4)
    assert f[0] == 1.0
    np.testing.assert_array_equal(orc.rope(x, f, 4, 2, 4, 0), x)
    r = orc.rope(x, f, 4, 2, 4, 1)
    np.testing.assert_allclose(r[0:2], [-np.sin(1.0), np.cos(1.0)], rtol=1e-6)
    # softmax of equal scores is uniform; argmax returns the FIRST maximum
    w = orc.attn_softmax(np.zeros((2, 4), np.float32))
    np.testing.assert_array_equal(w, np.full((2, 4), 0.25, np.float32))
    assert orc.argmax([1.0, 5.0, 5.0, 2.0]) == 1
    # SiLU(0) = 0, SiLU(large) ~ x
    np.testing.assert_allclose(orc.silu([0.0, 20.0, -20.0]), [0.0, 20.0, -20.0 / (1 + np.exp(20.0))],
                               rtol=1e-6, atol=1e-12)


def test_rope_freq_truncation_mode():
    full = orc.rope_freqs(128)
    trunc = orc.rope_freqs(128, n_valid=32)  # TransformerBlock.cs:66 uploads 32 entries only
    np.testing.assert_array_equal(full[:32], trunc[:32])
    assert (trunc[32:] == 0).all() and (full[32:] > 0).all()
    np.testing.assert_allclose(full, npo.rope_freqs(128), rtol=2e-6)


@pytest.mark.parametrize("N,K", [(64, 256), (96, 3072), (33, 8192)])
def test_gemv_vs_fp64(N, K):
    W = (0.02 * RNG.standard_normal((N, K))).astype(np.float16)
    x = RNG.standard_normal(K).astype(np.float32)
    y = orc.gemv(W.astype(np.float32), x)
    y16 = orc.gemv_f16w(W, x)
    np.testing.assert_array_equal(y, y16)  # widening is exact: same operands, same order
    ref = W.astype(np.float64) @ x.astype(np.float64)
    bound = 1e-6 * np.sqrt(K) * np.abs(W.astype(np.float64)) @ np.abs(x.astype(np.float64))
    assert (np.abs(y - ref) <= bound + 1e-7).all()


def test_attention_chain_vs_fp64():
    H, Hkv, D, C, S = 6, 2, 32, 40, 37
    q = RNG.standard_normal(H * D).astype(np.float32)
    Kc = RNG.standard_normal((C, Hkv * D)).astype(np.float32)
    Vc = RNG.standard_normal((C, Hkv * D)).astype(np.float32)
    s = orc.attn_scores(q, Kc, H, Hkv, D, S)
    w = orc.attn_softmax(s)
    o = orc.attn_wsum(w, Vc, H, Hkv, D, S)
    np.testing.assert_allclose(w.sum(axis=1), 1.0, rtol=1e-5)
    np.testing.assert_allclose(o, npo.attention(q, Kc, Vc, H, Hkv, D, S), rtol=2e-5, atol=2e-6)


def test_rmsnorm_rope_silu_vs_fp64():
    x = RNG.standard_normal(3072).astype(np.float32)
    g = (1 + 0.1 * RNG.standard_normal(3072)).astype(np.float32)
    np.testing.assert_allclose(orc.rmsnorm(x, g, 1e-5), npo.rmsnorm(x, g, 1e-5), rtol=2e-6)
    f = orc.rope_freqs(128)
    for pos in (0, 1, 17, 639):
        got = orc.rope(x, f, 128, 24, 128, pos)
        want = npo.rope(x, f, 128, 24, 128, pos)  # same fp32 table: isolates the rotation
        np.testing.assert_allclose(got, want, rtol=0, atol=3e-5 * (1 + pos / 64))
    np.testing.assert_allclose(orc.silu(x), npo.silu(x), rtol=2e-6, atol=1e-7)


def test_kquant_codecs():
    w = (0.02 * RNG.standard_normal(256 * 40)).astype(np.float32)
    b4 = orc.quantize_q4k(w)
    b6 = orc.quantize_q6k(w)
    assert b4.size == 40 * 144 and b6.size == 40 * 210
    d4, d6 = orc.dequant_q4k(b4, w.size), orc.dequant_q6k(b6, w.size)
    # decode side: C restatement == independent NumPy restatement of the block layout
    np.testing.assert_allclose(d4, npo.q4k_dequant(b4, w.size), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(d6, npo.q6k_dequant(b6, w.size), rtol=1e-6, atol=1e-9)
    # decode of RANDOM bytes (every scale/min bit pattern), finite halfs forced
    rb = RNG.integers(0, 256, 64 * 144, dtype=np.uint8).reshape(64, 144)
    rb[:, 1] &= 0x3B; rb[:, 3] &= 0x3B
    np.testing.assert_allclose(orc.dequant_q4k(rb, 64 * 256), npo.q4k_dequant(rb, 64 * 256), rtol=1e-6, atol=1e-6)
    rb = RNG.integers(0, 256, 64 * 210, dtype=np.uint8).reshape(64, 210)
    rb[:, 209] &= 0x3B
    np.testing.assert_allclose(orc.dequant_q6k(rb, 64 * 256), npo.q6k_dequant(rb, 64 * 256), rtol=1e-6, atol=1e-6)
    # round trip error of the build's own quantisers (4.5 / 6.56 bits per weight)
    assert np.abs(d4 - w).max() < 0.02 * 4 / 15 and np.sqrt(np.mean((d4 - w) ** 2)) < 2.5e-3
    assert np.abs(d6 - w).max() < 0.02 * 5 / 31 and np.sqrt(np.mean((d6 - w) ** 2)) < 6e-4


@pytest.mark.parametrize("dims", [synth.TINY, synth.TINY_D128])
def test_whole_model_vs_fp64(dims):
    w = synth.make_weights(dims, seed=5, std=0.05)
    desc = orc.LlamaDesc(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, C=16)
    m = orc.OracleLlama(desc, w)
    ref = npo.NpLlama(desc, w)
    toks = synth.make_tokens(dims, 6)
    for t in toks:
        lg = m.step(int(t))
        want = ref.step(int(t))
        np.testing.assert_allclose(lg, want, rtol=0, atol=2e-4 * np.abs(want).max())
        assert orc.argmax(lg) == int(np.argmax(want))
    # fp16-stored and fp32-widened weights give bit-identical logits
    m32 = orc.OracleLlama(desc, {k: v.astype(np.float32) for k, v in w.items()})
    m.reset()
    for t in toks[:2]:
        np.testing.assert_array_equal(m.step(int(t)), m32.step(int(t)))
    with pytest.raises(RuntimeError):
        for _ in range(20):
            m.step(1)


def test_layer_slices_compose():
    """A pipeline stage = a contiguous slice of the block loop (LlamaModel.cs:118-121): running
    [0,1) then [1,L) on the hidden state equals the whole step, bit for bit."""
    dims = synth.TINY_D128
    w = synth.make_weights(dims, seed=6, std=0.05)
    desc = orc.LlamaDesc(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, C=8)
    a, b = orc.OracleLlama(desc, w), orc.OracleLlama(desc, w)
    for t in (3, 9, 27):
        a.step(t, want_logits=False)
        h = w["token_embd.weight"][t].astype(np.float32)
        h = b.layers(h, 0, 1)
        h = b.layers(h, 1, dims.L)
        b.advance()
        np.testing.assert_array_equal(h, a.hidden())


def _topp_fp64(values, temperature, topP, topK, rand):
    """Independent evaluation of SamplingUtils.TopP (SamplingUtils.cs:5-33) in NumPy: float32 element arithmetic as the C#
    code has it, double accumulation where Enumerable.Sum is used."""
    scaled = values.astype(np.float32) / np.float32(temperature)
    e = np.exp(scaled - scaled.max()).astype(np.float32)
    probs = e / np.float32(e.sum(dtype=np.float64))
    order = np.argsort(-probs, kind="stable")[:topK]
    cum, keep = np.float32(0), 0
    for i in order:
        cum = np.float32(cum + probs[i])
        keep += 1
        if cum >= np.float32(topP):
            break
    kept = order[:keep]
    total = np.float32(probs[kept].sum(dtype=np.float64))
    running = np.float32(0)
    for i in kept:
        running = np.float32(running + probs[i] / total)
        if np.float32(rand) < running:
            return int(i), order, probs[order], keep
    return int(kept[-1]), order, probs[order], keep


@pytest.mark.parametrize("n,seed", [(1000, 1), (128256, 2), (50, 3)])
def test_topp_restatement(n, seed):
    """orc_topp (the restatement of the reference's default sampler) against an independent NumPy evaluation and against the
    product's host-side sampler; stable order of equal probabilities; the draw as an argument."""
    from nfai_amd.llama_model import SamplingUtils
    r = np.random.Generator(np.random.PCG64(seed))
    v = (3.0 * r.standard_normal(n)).astype(np.float32)
    top = v.max()
    v[[min(n - 1, 41), 7]] = top + 1.0  # two equal maxima: index 7 first (stable OrderByDescending, SamplingUtils.cs:9-12)
    k = min(40, n)
    for rand in (0.0, 0.05, 0.3, 0.5, 0.77, 0.949, 0.95, 0.999999):
        tok, ids, probs, kept = orc.topp(v, 0.5, 0.95, k, rand)
        tok2, ids2, probs2, kept2 = _topp_fp64(v, 0.5, 0.95, k, rand)
        assert tok == tok2 and kept == kept2
        np.testing.assert_array_equal(ids, ids2.astype(np.uint32))
        np.testing.assert_allclose(probs, probs2, rtol=2e-6)
        assert ids[0] == 7 and ids[1] == min(n - 1, 41)
        assert SamplingUtils.TopPFromCandidates(ids, probs, 0.95, rand=rand) == tok   # the host half the product keeps
    # temperature 1, a flat distribution: the nucleus needs all topK candidates
    flat = np.zeros(n, np.float32)
    tok, ids, probs, kept = orc.topp(flat, 1.0, 0.95, k, 0.5)
    assert kept == k and list(ids) == list(range(k)) and tok == _topp_fp64(flat, 1.0, 0.95, k, 0.5)[0] and tok in (19, 20)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="sanitizer builds run on the CPU container only (never on the GPU box)")
def test_oracle_under_address_and_ub_sanitizers():
    """oracle/san_driver.c drives every exported function of the oracle at ragged sizes on exactly-sized heap buffers under
    -fsanitize=address,undefined -fno-sanitize-recover=all (the reference runs with its validation layer always on,
    VulkanHelper.cs:14-17): any out-of-bounds or undefined operation aborts the child."""
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    subprocess.check_call(["make", "-C", here, "-B", "san_driver"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(here, "san_driver")], capture_output=True, text=True, timeout=300,
                       env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert r.returncode == 0 and "san_driver: ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]


def test_topp_from_candidates_vectorised_equals_the_loop_and_the_oracle():
    """The host half of SamplingUtils.TopP (SamplingUtils.cs:14-31) in its two Python forms — np.cumsum and one statement per line — and
    the oracle's restatement must draw the same token for every `rand`, also when the nucleus is cut at the first or at no element."""
    from nfai_amd.llama_model import SamplingUtils
    r = np.random.Generator(np.random.PCG64(77))
    for trial in range(60):
        n = int(r.integers(45, 400))
        logits = (r.standard_normal(n) * float(r.choice([0.3, 1.0, 4.0, 12.0]))).astype(np.float32)
        for rand in (0.0, 1e-7, 0.013, 0.25, 0.5, 0.77, 0.949, 0.951, 0.9999999):
            want, ids, probs, _ = orc.topp(logits, 0.5, 0.95, 40, rand)
            a = SamplingUtils.TopPFromCandidates(ids, probs, 0.95, rand=rand)
            b = SamplingUtils.TopPFromCandidatesLoop(ids, probs, 0.95, rand=rand)
            assert a == b == want, (trial, rand, a, b, want)


@pytest.mark.parametrize("dims", [synth.TINY, synth.TINY_D128], ids=lambda d: d.name)
def test_whole_model_vs_hf_transformers_llama(dims):
    """An implementation of the architecture that shares no code with this repository or with the reference: Hugging Face
    transformers' LlamaForCausalLM (fp32, eager attention, CPU) on the same synthetic weights.  GGUF files store Wq / Wk with the rows
    of a head permuted so that the rotation of NEIGHBOURING pairs (2i, 2i+1) — what RoPEShader.cs:238-271 and ggml do — equals
    transformers' rotate-half convention (i, i + D/2) on the original rows; the inverse permutation is applied here.  Every position's
    logits of one causal forward pass must equal the oracle's token-by-token logits (LlamaModel.cs:116-125) to fp32 summation noise.
    This does not pin the oracle to the REFERENCE (which holds no vectors: parity stays "unpinned") — it pins it to the
    model the reference implements: RMSNorm, GQA head mapping, 1/sqrt(D), the causal softmax, up * silu(gate), tied / untied lm_head."""
    torch = pytest.importorskip("torch")
    tf = pytest.importorskip("transformers")
    d = dims
    w = synth.make_weights(d, seed=5, std=0.05)
    cfg = tf.LlamaConfig(vocab_size=d.V, hidden_size=d.E, intermediate_size=d.F, num_hidden_layers=d.L, num_attention_heads=d.H,
                         num_key_value_heads=d.Hkv, head_dim=d.D, rms_norm_eps=1e-5, rope_theta=500000.0, tie_word_embeddings=d.tied,
                         attention_bias=False, mlp_bias=False, max_position_embeddings=64, attn_implementation="eager")
    hf = tf.LlamaForCausalLM(cfg).to(torch.float32).eval()

    def rotate_half_rows(wg, n_heads):  # GGUF row order (pairs interleaved) -> transformers' (first halves, then second halves)
        wg = wg.astype(np.float32).reshape(n_heads, d.D // 2, 2, -1)
        return np.concatenate([wg[:, :, 0, :], wg[:, :, 1, :]], axis=1).reshape(n_heads * d.D, -1)

    f32 = lambda a: a.astype(np.float32)  # noqa: E731
    sd = {"model.embed_tokens.weight": f32(w["token_embd.weight"]), "model.norm.weight": w["output_norm.weight"],
          "lm_head.weight": f32(w["token_embd.weight"] if d.tied else w["output.weight"])}
    for l in range(d.L):
        b, h = f"blk.{l}.", f"model.layers.{l}."
        sd[h + "input_layernorm.weight"] = w[b + "attn_norm.weight"]
        sd[h + "post_attention_layernorm.weight"] = w[b + "ffn_norm.weight"]
        sd[h + "self_attn.q_proj.weight"] = rotate_half_rows(w[b + "attn_q.weight"], d.H)
        sd[h + "self_attn.k_proj.weight"] = rotate_half_rows(w[b + "attn_k.weight"], d.Hkv)
        sd[h + "self_attn.v_proj.weight"] = f32(w[b + "attn_v.weight"])
        sd[h + "self_attn.o_proj.weight"] = f32(w[b + "attn_output.weight"])
        sd[h + "mlp.gate_proj.weight"] = f32(w[b + "ffn_gate.weight"])
        sd[h + "mlp.up_proj.weight"] = f32(w[b + "ffn_up.weight"])
        sd[h + "mlp.down_proj.weight"] = f32(w[b + "ffn_down.weight"])
    res = hf.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=False)
    assert not res.missing_keys and not res.unexpected_keys, res
    toks = synth.make_tokens(d, 24, seed=3)
    with torch.no_grad():
        want = hf(torch.from_numpy(toks.astype(np.int64))[None]).logits[0].numpy()
    ref = orc.OracleLlama(orc.LlamaDesc(E=d.E, L=d.L, H=d.H, Hkv=d.Hkv, D=d.D, F=d.F, V=d.V, C=32), w)
    for i, t in enumerate(toks):
        got = ref.step(int(t))
        assert np.abs(got - want[i]).max() <= 1e-4 * max(1.0, float(np.abs(want[i]).max())), i   # observed 1.2e-5 at |logit| <= 5.5
        assert int(np.argmax(got)) == int(np.argmax(want[i])), i
