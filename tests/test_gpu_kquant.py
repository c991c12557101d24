"""K-quant (Q4_K / Q6_K) decode path on the GPU against the oracle.

The reference cannot load these tensor types (NFAI.GGUF/Parser.cs:111-114 throws), so parity is
UNPINNED by the reference: the oracle here is "dequantise the blocks (ggml layout, restated in
oracle/nfai_oracle.c) then the reference's fp32 GEMV".  Blocks come from the build's own quantisers
(valid blocks with every field exercised; random-byte blocks are covered on the decode side in
tests/test_oracle.py)."""
import numpy as np
import pytest

import oracle as orc
from nfai_amd import synth

pytestmark = pytest.mark.gpu

Q4_K, Q6_K = 12, 14


@pytest.fixture(scope="module")
def mgr():
    from nfai_amd.hip import HipBufferManager
    m = HipBufferManager(0)
    yield m
    m.Dispose()


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def quantize(W, qt):
    """W [N][K] fp32 -> (raw block bytes, dequantised fp32 [N][K])."""
    N, K = W.shape
    if qt == Q4_K:
        b = orc.quantize_q4k(W)
        return b, orc.dequant_q4k(b, N * K).reshape(N, K)
    b = orc.quantize_q6k(W)
    return b, orc.dequant_q6k(b, N * K).reshape(N, K)


def tol(Wd, x):
    s = np.abs(Wd.astype(np.float64)) @ np.abs(x.astype(np.float64))
    return 2e-6 * np.sqrt(Wd.shape[1] / 256.0) * s + 1e-6


@pytest.mark.parametrize("qt", [Q4_K, Q6_K])
@pytest.mark.parametrize("N,K", [(2048, 2048), (1024, 3072), (512, 8192), (96, 256), (40, 768), (300, 14336), (7, 4096), (64, 28672), (304, 14336)])
def test_gemv_kquant(mgr, qt, N, K):
    from nfai_amd.shaders import MatrixMultiplyShader
    r = rng(N + K + qt)
    W = (0.02 * r.standard_normal((N, K))).astype(np.float32)
    raw, Wd = quantize(W, qt)
    x = r.standard_normal(K).astype(np.float32)
    op = MatrixMultiplyShader(mgr, 1, K, N, None)
    op.GetWeightProperty().set(raw, qt, N, K)
    op.GetInputProperty().SetValue(x)
    op.Compute()
    ref = orc.gemv(Wd, x)
    err = np.abs(op.GetOutputs() - ref)
    assert (err <= tol(Wd, x)).all(), (err.max(), tol(Wd, x).min())


@pytest.mark.parametrize("qt", [Q4_K, Q6_K])
def test_embed_kquant(mgr, qt):
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    r = rng(qt)
    V, E = 300, 768
    W = (0.05 * r.standard_normal((V, E))).astype(np.float32)
    raw, Wd = quantize(W, qt)
    tab = mgr.UploadWeight(qt, raw, V, E)
    tok, y = ShaderProperty(mgr, 1, np.uint32), ShaderProperty(mgr, E)
    for t in (0, 1, 299, 123):
        tok.SetValue(np.array([t], np.uint32))
        call("nfai_hip_embed", mgr.handle, tab.handle, qt, tok.handle, y.handle, E)
        np.testing.assert_allclose(y.GetValue(), Wd[t], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("qt", [Q4_K, Q6_K])
@pytest.mark.parametrize("V,E", [(128256, 768), (4000, 3072), (48, 256)])
def test_lmhead_argmax_kquant(mgr, qt, V, E):
    """RMSNorm + K-quant lm_head + ArgMax in one launch: logits against the dequantised GEMV, index = first maximum of the
    launch's own logits; equal maxima (copied rows) in different workgroups resolve to the lowest index."""
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    r = rng(V + E + qt)
    rows = min(V, 2048)
    W = np.tile((0.02 * r.standard_normal((rows, E))).astype(np.float32), ((V + rows - 1) // rows, 1))[:V].copy()
    W *= (1 + 0.01 * r.standard_normal((V, 1))).astype(np.float32)
    x = r.standard_normal(E).astype(np.float32)
    g = (1 + 0.1 * r.standard_normal(E)).astype(np.float32)
    xn = orc.rmsnorm(x, g, 1e-5)
    dup = (V - 1, 35, V // 2 + 1)
    for j in dup:
        W[j] = np.sign(xn) * 0.06
    raw, Wd = quantize(W, qt)
    tab = mgr.UploadWeight(qt, raw, V, E)
    px, pg, pl, pi = ShaderProperty(mgr, E), ShaderProperty(mgr, E), ShaderProperty(mgr, V), ShaderProperty(mgr, 1, np.uint32)
    px.SetValue(x); pg.SetValue(g)
    for _ in range(2):
        call("nfai_hip_lmhead_argmax", mgr.handle, tab.handle, qt, px.handle, pg.handle, 1e-5, pl.handle, pi.handle, V, E)
        lg = pl.GetValue()
        sl = slice(0, min(V, 2500))
        assert (np.abs(lg[sl] - orc.gemv(Wd[sl], xn)) <= tol(Wd[sl], xn)).all()
        assert int(pi.GetValue()[0]) == int(np.argmax(lg)) == min(dup)


@pytest.mark.parametrize("qt", [Q4_K, Q6_K])
@pytest.mark.parametrize("E,F", [(3072, 8192), (2048, 8192), (256, 512)])
def test_gateup_and_residual_kquant(mgr, qt, E, F):
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    r = rng(E + F + qt)
    Wg = (0.02 * r.standard_normal((F, E))).astype(np.float32)
    Wu = (0.02 * r.standard_normal((F, E))).astype(np.float32)
    Wd_ = (0.02 * r.standard_normal((E, F))).astype(np.float32)
    (rg, dg), (ru, du), (rd, dd) = quantize(Wg, qt), quantize(Wu, qt), quantize(Wd_, qt)
    x = r.standard_normal(E).astype(np.float32)
    g = (1 + 0.1 * r.standard_normal(E)).astype(np.float32)
    pg_, pu, pd = mgr.UploadWeight(qt, rg, F, E), mgr.UploadWeight(qt, ru, F, E), mgr.UploadWeight(qt, rd, E, F)
    px, pg, pa, py = ShaderProperty(mgr, E), ShaderProperty(mgr, E), ShaderProperty(mgr, F), ShaderProperty(mgr, E)
    px.SetValue(x); pg.SetValue(g)
    call("nfai_hip_gemv_gateup_silu", mgr.handle, pg_.handle, pu.handle, qt, px.handle, pg.handle, 1e-5, pa.handle, F, E)
    xn = orc.rmsnorm(x, g, 1e-5)
    act = orc.mul(orc.gemv(du, xn), orc.silu(orc.gemv(dg, xn)))
    np.testing.assert_allclose(pa.GetValue(), act, rtol=1e-4, atol=2e-5)
    call("nfai_hip_gemv_fused", mgr.handle, pd.handle, qt, pa.handle, 0, 0.0, px.handle, py.handle, E, F)
    want = orc.add(x, orc.gemv(dd, pa.GetValue()))
    assert (np.abs(py.GetValue() - want) <= tol(dd, pa.GetValue()) + 1e-5).all()


@pytest.mark.parametrize("vq", [Q4_K, Q6_K])
def test_qkv_rope_kquant_mixed_v(mgr, vq):
    """q, k in Q4_K with v in Q4_K or Q6_K (the Q4_K_M mix): through the model path, which splits the
    launch when the encodings differ."""
    from nfai_amd.llama_model import LlamaModel, QuantTensor
    dims = synth.TINY_D128
    w = synth.make_weights(dims, seed=61, std=0.05)
    wq, wref = {}, {}
    for name, a in w.items():
        if a.ndim == 1:
            wq[name] = a
            wref[name] = a
            continue
        qt = Q6_K if (name.endswith(("attn_v.weight", "ffn_down.weight")) and vq == Q6_K) or name.startswith(("token_embd", "output.")) else Q4_K
        raw, deq = quantize(a.astype(np.float32), qt)
        wq[name] = QuantTensor(raw, qt, a.shape)
        wref[name] = deq
    md = synth.make_metadata(dims)
    dd = dict(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, eps=1e-5, rope_dims=dims.D, rope_base=500000.0)
    m = LlamaModel(mgr, md, wq, 40, dims=dd)
    mu = LlamaModel(mgr, md, wq, 40, dims=dd, unfused=True)
    ref = orc.OracleLlama(orc.LlamaDesc(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, C=40), wref)
    for i, t in enumerate(synth.make_tokens(dims, 32, seed=9)):
        lg, am = m.Step(int(t))
        lu, _ = mu.Step(int(t))
        want = ref.step(int(t))
        scale = max(1.0, float(np.abs(want).max()))
        assert np.abs(lg - want).max() <= 5e-4 * scale, (i, np.abs(lg - want).max())
        assert np.abs(lu - want).max() <= 5e-4 * scale
        assert am == orc.argmax(want)
    total, _ = m.BytesPerToken(0)
    assert total < sum(a.nbytes for a in w.values() if a.ndim == 2) * 0.45  # 4.5-6.6 bits instead of 16
    m.Dispose()
    mu.Dispose()


@pytest.mark.parametrize("n,chunk", [(70, 64), (40, 128)], ids=["chunked", "one-chunk"])
def test_prefill_mfma_kquant(mgr, n, chunk):
    """Q4_K_M-style model (Q4_K blocks, attn_v / ffn_down / embeddings in Q6_K) through the batched MFMA prefill: each
    block's matrices are widened to fp16 scratch and go through the fp16 GEMMs.  Oracle = token-by-token fp32 on the
    dequantised weights; stated fp16 tolerance as for the fp16 prefill (tests/test_gpu_model.py): logits
    max|d| <= 2e-2 * max(1, max|logit|), same argmax, decode continues from the prefilled cache."""
    from nfai_amd.llama_model import LlamaModel, QuantTensor
    dims = synth.TINY_D128
    w = synth.make_weights(dims, seed=67, std=0.05)
    wq, wref = {}, {}
    for name, a in w.items():
        if a.ndim == 1:
            wq[name] = a
            wref[name] = a
            continue
        qt = Q6_K if name.endswith(("attn_v.weight", "ffn_down.weight")) or name.startswith(("token_embd", "output.")) else Q4_K
        raw, deq = quantize(a.astype(np.float32), qt)
        wq[name] = QuantTensor(raw, qt, a.shape)
        wref[name] = deq
    dd = dict(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, eps=1e-5, rope_dims=dims.D, rope_base=500000.0)
    m = LlamaModel(mgr, synth.make_metadata(dims), wq, 160, dims=dd, max_batch=chunk)
    ref = orc.OracleLlama(orc.LlamaDesc(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, C=160), wref)
    toks = synth.make_tokens(dims, n, seed=21)
    want = None
    for t in toks:
        want = ref.step(int(t))
    got = m.Prefill(toks)
    assert m.Pos == n
    tol5 = 2e-2 * max(1.0, float(np.abs(want).max()))
    assert np.abs(got - want).max() <= tol5, np.abs(got - want).max()
    assert int(np.argmax(got)) == orc.argmax(want)
    np.testing.assert_allclose(m.ReadKV(dims.L - 1, False, n - 1), ref.kcache(dims.L - 1)[n - 1], rtol=0, atol=2e-2)
    tok = orc.argmax(want)
    for _ in range(6):
        lg, am = m.Step(tok)
        wl = ref.step(tok)
        assert np.abs(lg - wl).max() <= tol5
        tok = orc.argmax(wl)
    m.Dispose()


def test_prefill_kept_fp16_copies_match_the_per_block_scratch(mgr, monkeypatch):
    """K-quant prefill: the blocks' matrices are widened to fp16 once and kept (the default while they fit a quarter of the HBM) or
    re-widened into one block's scratch for every block and chunk (NFAI_PREFILL_WIDE_ALL=0, read when the scratch is allocated).
    Same kernels on the same bytes: logits and K / V rows must be IDENTICAL — for the first prefill (which widens), for a second one
    that reuses the kept copies, and after a weight was replaced (the copies are re-made)."""
    from nfai_amd._lib import call
    from nfai_amd.llama_model import LlamaModel, QuantTensor
    dims = synth.TINY_D128
    w = synth.make_weights(dims, seed=68, std=0.05)
    wq = {}
    for name, a in w.items():
        if a.ndim == 1:
            wq[name] = a
            continue
        qt = Q6_K if name.endswith(("attn_v.weight", "ffn_down.weight")) or name.startswith(("token_embd", "output.")) else Q4_K
        wq[name] = QuantTensor(quantize(a.astype(np.float32), qt)[0], qt, a.shape)
    dd = dict(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, eps=1e-5, rope_dims=dims.D, rope_base=500000.0)
    toks = synth.make_tokens(dims, 100, seed=22)
    other = "blk.1.ffn_gate.weight"
    repl = QuantTensor(quantize((0.5 * w[other]).astype(np.float32), Q4_K)[0], Q4_K, w[other].shape)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NFAI_PREFILL_WIDE_ALL", mode)
        m = LlamaModel(mgr, synth.make_metadata(dims), wq, 128, dims=dd, max_batch=64)
        a = m.Prefill(toks)
        kv_a = m.ReadKV(dims.L - 1, True, 99)
        m.Reset()
        b = m.Prefill(toks)            # mode 1: every block's copy is reused
        assert np.array_equal(a, b) and np.array_equal(kv_a, m.ReadKV(dims.L - 1, True, 99))
        m.SetTensor(other, repl)       # a replaced weight must reach the prefill GEMMs
        call("nfai_hip_llama_finalize", m.handle)
        m.Reset()
        c = m.Prefill(toks)
        assert not np.array_equal(a, c)
        outs[mode] = (a, kv_a, c)
        m.Dispose()
    for x, y in zip(outs["1"], outs["0"]):
        assert np.array_equal(x, y)


def test_prefill_fused_dequant_in_child_process():
    """The dequant-in-LDS prefill (k_gemm_kq, NFAI_PREFILL_FUSED=1) at model level.  The switch is read once per process, so the
    same Q4_K_M-style prefill test runs again in a child process with the variable set; the child's interpreter loads the
    library itself (nothing is inherited but the environment)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, NFAI_PREFILL_FUSED="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_kquant.py"), "-m", "gpu", "-x", "-q",
                        "-k", "test_prefill_mfma_kquant", "-p", "no:cacheprovider"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "2 passed" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("qt", [Q4_K, Q6_K])
@pytest.mark.parametrize("M,N,K,res", [(256, 512, 1024, True), (100, 64, 256, False), (512, 3072, 768, False)])
def test_gemm_kq_dequant_in_lds(mgr, qt, M, N, K, res):
    """The dequant-in-LDS GEMM (k_gemm_kq: quantised bytes -> registers -> fp16 tile in LDS -> MFMA) against fp64 NumPy on the
    oracle's dequantised weights ROUNDED TO fp16 (the kernel forms d*sc*q - dmin*m in fp32 and rounds to fp16 on the way into
    LDS, like the separate widening pass): with identical operands the error is fp32 summation-order noise.  Covers both tile
    configurations (N = 3072 at M = 512 takes 128x128), ragged M and the residual epilogue."""
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    r = rng(7 * M + N + K + qt)
    W = (0.02 * r.standard_normal((N, K))).astype(np.float32)
    raw, Wd = quantize(W, qt)
    A = r.standard_normal((M, K)).astype(np.float16)
    R = r.standard_normal((M, N)).astype(np.float32) if res else None
    wbuf = mgr.UploadWeight(qt, raw, N, K)
    pa, pc = ShaderProperty(mgr, M * K, np.float16), ShaderProperty(mgr, M * N, np.float32)
    pa.SetValue(A.ravel())
    pr = None
    if res:
        pr = ShaderProperty(mgr, M * N, np.float32)
        pr.SetValue(R.ravel())
    call("nfai_hip_gemm_kq", mgr.handle, pa.handle, wbuf.handle, qt, pr.handle if res else 0, pc.handle, M, N, K)
    got = pc.GetValue().reshape(M, N)
    W16 = Wd.astype(np.float16).astype(np.float64)
    want = A.astype(np.float64) @ W16.T + (R.astype(np.float64) if res else 0.0)
    scale = float(np.abs(A.astype(np.float64)).mean() * np.abs(W16).mean() * K)
    assert np.abs(got - want).max() <= 2e-6 * np.sqrt(K) * scale + 1e-5, np.abs(got - want).max()


@pytest.mark.parametrize("qt", [Q4_K, Q6_K])
@pytest.mark.parametrize("xscale", [0.0, 1e-30, 1e-6, 3e4], ids=["zero", "tiny", "small", "large"])
def test_gemv_kquant_activation_range(mgr, qt, xscale):
    """The int8-MFMA GEMV turns every 256-element super-block of x into fixed point with its own power-of-two scale: the result
    must track the fp32 oracle over the whole dynamic range (all-zero input, magnitudes far below and above 1, and one
    super-block 10^6 times larger than its neighbours)."""
    from nfai_amd.shaders import MatrixMultiplyShader
    N, K = 256, 2048
    r = rng(1000 + qt)
    W = (0.02 * r.standard_normal((N, K))).astype(np.float32)
    raw, Wd = quantize(W, qt)
    x = (xscale * r.standard_normal(K)).astype(np.float32)
    if xscale == 1e-6:
        x[256:512] *= 1e6  # one loud super-block beside quiet ones
    op = MatrixMultiplyShader(mgr, 1, K, N, None)
    op.GetWeightProperty().set(raw, qt, N, K)
    op.GetInputProperty().SetValue(x)
    op.Compute()
    got = op.GetOutputs()
    ref = orc.gemv(Wd, x)
    assert np.isfinite(got).all()
    assert (np.abs(got - ref) <= tol(Wd, x) + 1e-30).all(), (np.abs(got - ref).max(), tol(Wd, x).min())
