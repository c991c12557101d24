"""GPU parity of every operator of the HIP backend against the CPU oracle, through the C ABI.

Inputs are seeded; sizes are the BASELINE.json shapes where the oracle finishes in seconds
(GEMV rows are sub-sampled for the big matrices by slicing N, never K).  Tolerances are fp32
summation-order bounds: the HIP kernels add the same fp32 products in a different (tree) order.
"""
import numpy as np
import pytest

import oracle as orc
from oracle import np_oracle as npo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mgr():
    from nfai_amd.hip import HipBufferManager
    m = HipBufferManager(0)
    yield m
    m.Dispose()


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def gemv_tol(W, x):
    """|sum_k fl(w*x)| reordering bound: a few ulp of sum |w||x| times log-ish factor."""
    s = np.abs(W.astype(np.float64)) @ np.abs(x.astype(np.float64))
    return 4e-7 * np.sqrt(W.shape[1] / 64.0) * s + 1e-7


def test_device_is_gfx950(mgr):
    assert mgr.info.arch.decode().startswith("gfx950")
    assert mgr.info.wavefront_size == 64


# ---- MatrixMultiplyShader -------------------------------------------------------------------
@pytest.mark.parametrize("N,K", [
    (2048, 2048), (512, 2048), (3072, 3072), (1024, 3072), (1000, 8192),   # 1B / 3B layer shapes (N sub-sampled where large)
    (1024, 4096), (512, 14336),                                            # 8B
    (7, 256), (33, 264), (1, 8), (130, 1000), (257, 520),                  # ragged: K not a multiple of 512, tiny N
])
def test_gemv_f16(mgr, N, K):
    from nfai_amd.shaders import MatrixMultiplyShader
    r = rng(N * 131 + K)
    W = (0.02 * r.standard_normal((N, K))).astype(np.float16)
    x = r.standard_normal(K).astype(np.float32)
    op = MatrixMultiplyShader(mgr, 1, K, N, W)
    op.GetInputProperty().SetValue(x)
    op.Compute()
    y = op.GetOutputs()
    ref = orc.gemv_f16w(W, x)
    assert (np.abs(y - ref) <= gemv_tol(W, x)).all(), np.abs(y - ref).max()


def test_gemv_f32_weights(mgr):
    from nfai_amd.shaders import MatrixMultiplyShader
    r = rng(5)
    W = r.standard_normal((300, 768)).astype(np.float32)
    x = r.standard_normal(768).astype(np.float32)
    op = MatrixMultiplyShader(mgr, 1, 768, 300, W)
    op.GetInputProperty().SetValue(x)
    op.Compute()
    assert (np.abs(op.GetOutputs() - orc.gemv(W, x)) <= gemv_tol(W, x)).all()


def test_gemv_lm_head_full_size(mgr):
    """128256 x 3072 fp16 (788 MB): every row against the oracle, plus linearity y(a+b) = y(a)+y(b)."""
    from nfai_amd.shaders import MatrixMultiplyShader
    r = rng(77)
    N, K = 128256, 3072
    W = (0.02 * r.standard_normal((N, K), dtype=np.float32)).astype(np.float16)
    a = r.standard_normal(K).astype(np.float32)
    b = r.standard_normal(K).astype(np.float32)
    op = MatrixMultiplyShader(mgr, 1, K, N, W)
    outs = []
    for v in (a, b, a + b):
        op.GetInputProperty().SetValue(v)
        op.Compute()
        outs.append(op.GetOutputs())
    ref = orc.gemv_f16w(W, a)
    assert (np.abs(outs[0] - ref) <= gemv_tol(W, a)).all()
    np.testing.assert_allclose(outs[2], outs[0] + outs[1], rtol=0, atol=2e-5)


def test_gemv_cached_rows_and_overflow(mgr):
    """The KV-cached variant writes row currentCacheSize and advances (MatrixMultiplyShader.cs:247-252,
    :286-287); past the capacity the reference writes out of bounds, this backend raises."""
    from nfai_amd.shaders import MatrixMultiplyShader
    from nfai_amd._lib import KVCacheFull
    r = rng(9)
    W = (0.05 * r.standard_normal((64, 256))).astype(np.float16)
    op = MatrixMultiplyShader(mgr, 1, 256, 64, W, contextSize=3)
    rows = []
    for i in range(3):
        x = r.standard_normal(256).astype(np.float32)
        op.GetInputProperty().SetValue(x)
        op.Compute()
        rows.append(orc.gemv_f16w(W, x))
    got = op.GetOutputs().reshape(3, 64)
    np.testing.assert_allclose(got, np.stack(rows), rtol=0, atol=1e-5)
    with pytest.raises(KVCacheFull):
        op.Compute()


# ---- small ops ------------------------------------------------------------------------------
@pytest.mark.parametrize("E", [256, 2048, 3072, 4096, 1000])
def test_rmsnorm(mgr, E):
    from nfai_amd.shaders import RMSNormShader
    r = rng(E)
    x = r.standard_normal(E).astype(np.float32) * 3
    g = (1 + 0.1 * r.standard_normal(E)).astype(np.float32)
    op = RMSNormShader(mgr, E, g, 1e-5)
    op.GetInputProperty().SetValue(x)
    op.Compute()
    np.testing.assert_allclose(op.GetOutputs(), orc.rmsnorm(x, g, 1e-5), rtol=3e-6, atol=1e-7)


@pytest.mark.parametrize("H,D,pos", [(32, 64, 0), (32, 64, 1), (24, 128, 17), (24, 128, 639), (8, 128, 511)])
@pytest.mark.parametrize("nfreq", [None, 32])
def test_rope(mgr, H, D, pos, nfreq):
    from nfai_amd.shaders import RoPEShader
    r = rng(H * D + pos)
    x = r.standard_normal(H * D).astype(np.float32)
    freqs = orc.rope_freqs(D, 500000.0, nfreq)
    table = freqs if nfreq is None else freqs[:nfreq]  # the reference uploads 32 entries (TransformerBlock.cs:66)
    op = RoPEShader(mgr, H * D, H * D, table, D, H)
    op.GetInputProperty().SetValue(x)
    op.Compute(pos)
    want = orc.rope(x, freqs, D, H, D, pos)
    # device sinf/cosf vs libm at |theta| up to 639 rad: a few ulp of the angle
    np.testing.assert_allclose(op.GetOutputs(), want, rtol=0, atol=2e-6 * (1 + pos / 8))
    if nfreq == 32 and D == 128:  # reference defect reproduced on request: dims 64..127 not rotated
        got = op.GetOutputs().reshape(H, D)
        np.testing.assert_array_equal(got[:, 64:], x.reshape(H, D)[:, 64:])


def test_rope_in_place_on_cache_row(mgr):
    from nfai_amd.shaders import RoPEShader
    r = rng(3)
    Hkv, D, C, pos = 8, 64, 16, 5
    cache = r.standard_normal(C * Hkv * D).astype(np.float32)
    op = RoPEShader(mgr, C * Hkv * D, C * Hkv * D, orc.rope_freqs(D), D, Hkv, C)
    op.GetOutputProperty().BindShaderProprty(op.GetInputProperty())
    op.GetInputProperty().SetValue(cache)
    op.Compute(pos)
    got = op.GetInputProperty().GetValue().reshape(C, Hkv * D)
    want = cache.reshape(C, Hkv * D).copy()
    want[pos] = orc.rope(want[pos], orc.rope_freqs(D), D, Hkv, D, pos)
    np.testing.assert_allclose(got, want, rtol=0, atol=3e-6)
    np.testing.assert_array_equal(np.delete(got, pos, 0), np.delete(want, pos, 0))  # other rows untouched


@pytest.mark.parametrize("H,Hkv,D,S,C", [(32, 8, 64, 1, 8), (32, 8, 64, 130, 256), (24, 8, 128, 77, 128), (24, 8, 128, 640, 640)])
def test_attention_three_stage_chain(mgr, H, Hkv, D, S, C):
    from nfai_amd.shaders import (AttentionScoreCalculationShader, AttentionSoftmaxShader,
                                  AttentionWeightedValueSumShader)
    r = rng(H + S)
    q = r.standard_normal(H * D).astype(np.float32)
    Kc = r.standard_normal((C, Hkv * D)).astype(np.float32)
    Vc = r.standard_normal((C, Hkv * D)).astype(np.float32)
    sc = AttentionScoreCalculationShader(mgr, H, Hkv, C, D)
    sm = AttentionSoftmaxShader(mgr, H, C, D, 1e-5)
    ws = AttentionWeightedValueSumShader(mgr, H, Hkv, C, D)
    sm.GetInputProperty().BindShaderProprty(sc.GetAttentionScoresProperty())
    ws.GetAttentionWeights().BindShaderProprty(sm.GetAttentionWeightsProperty())
    sc.GetQueryVectorsProperty().SetValue(q)
    sc.GetKeyCacheProperty().SetValue(Kc)
    ws.GetValueCache().SetValue(Vc)
    sc.ComputeAttention(S)
    s_ref = orc.attn_scores(q, Kc, H, Hkv, D, S)
    got_s = sc.GetAttentionScoresProperty().GetValue()[:H * S].reshape(H, S)  # packed with stride S (:204)
    np.testing.assert_allclose(got_s, s_ref, rtol=0, atol=2e-5)
    sm.ComputeSoftmax(S)
    w_ref = orc.attn_softmax(s_ref)
    np.testing.assert_allclose(sm.GetAttentionWeightsProperty().GetValue()[:H * S].reshape(H, S), w_ref, rtol=2e-5, atol=1e-8)
    ws.ComputeWeightedSum(S)
    o_ref = orc.attn_wsum(w_ref, Vc, H, Hkv, D, S)
    np.testing.assert_allclose(ws.GetAttentionOutputProperty().GetValue(), o_ref, rtol=0, atol=2e-5)


def test_silu_mul_add_embed_argmax(mgr):
    import ctypes as C
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    from nfai_amd.shaders import SiLUShader, ElementWiseMultiplicationShader, TokenEmbedShader
    r = rng(21)
    n = 8192
    a = (4 * r.standard_normal(n)).astype(np.float32)
    b = r.standard_normal(n).astype(np.float32)
    s = SiLUShader(mgr, n)
    s.GetInputProperty().SetValue(a)
    s.Compute()
    np.testing.assert_allclose(s.GetOutputProperty().GetValue(), orc.silu(a), rtol=3e-6, atol=1e-7)
    m = ElementWiseMultiplicationShader(mgr, n)
    m.GetInputA().SetValue(a)
    m.GetInputB().SetValue(b)
    m.Compute()
    np.testing.assert_array_equal(m.GetOutputProperty().GetValue(), orc.mul(a, b))
    pa, pb, py = (ShaderProperty(mgr, n) for _ in range(3))
    pa.SetValue(a)
    pb.SetValue(b)
    call("nfai_hip_add", mgr.handle, pa.handle, pb.handle, py.handle, n)
    np.testing.assert_array_equal(py.GetValue(), orc.add(a, b))
    emb = (r.standard_normal((1000, 256))).astype(np.float16)
    e = TokenEmbedShader(mgr, 1, 256, emb)
    e.Compute(999)
    np.testing.assert_array_equal(e.GetOutputs(), emb[999].astype(np.float32))
    # argmax: first maximum wins (SamplingUtils.cs:56), also with duplicates spread over blocks
    v = r.standard_normal(128256).astype(np.float32)
    v[[77, 90001, 128255]] = 9.5
    pv, pi = ShaderProperty(mgr, v.size), ShaderProperty(mgr, 1, np.uint32)
    for _ in range(3):  # the device ticket must re-arm between launches
        pv.SetValue(v)
        call("nfai_hip_argmax", mgr.handle, pv.handle, v.size, pi.handle)
        assert int(pi.GetValue()[0]) == 77 == orc.argmax(v)
    v[3] = 11.0
    pv.SetValue(v)
    call("nfai_hip_argmax", mgr.handle, pv.handle, v.size, pi.handle)
    assert int(pi.GetValue()[0]) == 3


# ---- output norm + lm_head + ArgMax in one launch (LlamaModel.cs:123-125 + SamplingUtils.cs:43-57) -------------------------------
@pytest.mark.parametrize("V,E,norm", [(128256, 256, True), (128256, 3072, True), (32000, 2048, False), (1000, 264, True), (17, 8, False)])
def test_lmhead_argmax_one_launch(mgr, V, E, norm):
    """logits as the plain GEMV gives them, and the FIRST index of the maximum (SamplingUtils.cs:55-56) also when equal maxima
    come out of different workgroups: rows 77, V // 2 + 3 and V - 1 of the table are copies of one row with the largest output."""
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    r = rng(V + E)
    rows = min(V, 4096)                                   # the oracle GEMV runs on a slice; the full table is tiled from it
    base = (0.02 * r.standard_normal((rows, E))).astype(np.float16)
    W = np.tile(base, ((V + rows - 1) // rows, 1))[:V].copy()
    W += (1e-3 * r.standard_normal((V, 1))).astype(np.float16)   # rows differ again
    x = r.standard_normal(E).astype(np.float32)
    g = (1 + 0.1 * r.standard_normal(E)).astype(np.float32)
    xn = orc.rmsnorm(x, g, 1e-5) if norm else x
    if V >= 1000:
        big = (np.sign(xn) * 0.05).astype(np.float16)     # a row whose output beats every other
        for j in (V - 1, 77, V // 2 + 3):
            W[j] = big
    tab = mgr.UploadWeight(1, W, V, E)
    px, pg, pl, pi = ShaderProperty(mgr, E), ShaderProperty(mgr, E), ShaderProperty(mgr, V), ShaderProperty(mgr, 1, np.uint32)
    px.SetValue(x)
    pg.SetValue(g)
    for _ in range(3):  # the ticket re-arms
        call("nfai_hip_lmhead_argmax", mgr.handle, tab.handle, 1, px.handle, pg.handle if norm else 0, 1e-5, pl.handle, pi.handle, V, E)
        lg = pl.GetValue()
        sl = slice(0, min(V, 3000))
        ref = orc.gemv_f16w(W[sl], xn)
        assert (np.abs(lg[sl] - ref) <= gemv_tol(W[sl].astype(np.float32), xn)).all()
        assert int(pi.GetValue()[0]) == int(np.argmax(lg)) == orc.argmax(lg)      # exact on the launch's own logits
        if V >= 1000:
            assert int(pi.GetValue()[0]) == 77


# ---- candidates of SamplingUtils.TopP on the device (SamplingUtils.cs:5-13) --------------------------------------------------
@pytest.mark.parametrize("n,k,temperature,seed", [(128256, 40, 0.5, 1), (128256, 64, 1.0, 2), (32000, 40, 0.5, 3), (1000, 40, 0.5, 4),
                                                  (40, 40, 0.7, 5), (65, 1, 0.5, 6), (300000, 40, 0.5, 7)])
def test_topk_candidates(mgr, n, k, temperature, seed):
    """nfai_hip_topk = scale by 1/temperature, softmax over all n, stable descending order, first k (SamplingUtils.cs:7-13):
    indices exact against the oracle's restatement (ties: lower index first, also across workgroups), probabilities to 1e-6
    relative (device expf and the order of the sum differ), and the token TopP then draws identical for a sweep of `rand`."""
    from nfai_amd import _lib
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    from nfai_amd.llama_model import SamplingUtils
    import ctypes as C
    r = rng(seed)
    v = (3.0 * r.standard_normal(n)).astype(np.float32)
    if n >= 1000:
        v[[n - 1, 77, n // 2 + 3]] = v.max() + 0.75       # equal maxima in three different workgroups: index order decides
        v[[5, n // 3, n // 3 + 1]] = np.sort(v)[-20]      # a tie inside the top-k
    pv = ShaderProperty(mgr, n)
    ids, probs = np.empty(k, np.uint32), np.empty(k, np.float32)
    for rep in range(3):  # the device ticket re-arms between launches
        pv.SetValue(v)
        call("nfai_hip_topk", mgr.handle, pv.handle, n, temperature, k, ids.ctypes.data_as(C.POINTER(C.c_uint32)),
             probs.ctypes.data_as(C.POINTER(C.c_float)))
        _, ids_ref, probs_ref, _ = orc.topp(v, temperature, 0.95, k, 0.0)
        np.testing.assert_array_equal(ids, ids_ref)
        np.testing.assert_allclose(probs, probs_ref, rtol=1e-6)
        for rand in (0.0, 0.01, 0.2, 0.5, 0.8, 0.94, 0.97, 0.999999):
            assert SamplingUtils.TopPFromCandidates(ids, probs, 0.95, rand=rand) == orc.topp(v, temperature, 0.95, k, rand)[0]
        v = np.roll(v, 12345 % n)
    for bad in ((0.0, k), (0.5, 0), (0.5, 65), (0.5, n + 1)):
        with pytest.raises(_lib.NfaiHipError):
            call("nfai_hip_topk", mgr.handle, pv.handle, n, bad[0], bad[1], ids.ctypes.data_as(C.POINTER(C.c_uint32)),
                 probs.ctypes.data_as(C.POINTER(C.c_float)))


# ---- fused operators vs the unfused oracle chain, each at its own scale ----------------------
@pytest.mark.parametrize("H,Hkv,D,S,C", [(32, 8, 64, 1, 64), (32, 8, 64, 15, 64), (32, 8, 64, 16, 64), (32, 8, 64, 17, 64),
                                         (32, 8, 64, 1024, 1024), (24, 8, 128, 5, 640), (24, 8, 128, 513, 640),
                                         (24, 8, 128, 640, 640), (32, 8, 128, 300, 4096), (4, 2, 64, 33, 40), (8, 8, 128, 100, 128),
                                         (24, 8, 128, 4001, 4096), (32, 8, 64, 8000, 8192), (24, 8, 128, 1500, 2048)])  # multi-iteration slices
@pytest.mark.parametrize("kv16", [False, True])
def test_attn_decode_fused(mgr, H, Hkv, D, S, C, kv16):
    import ctypes as Cc
    from nfai_amd import _lib
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    r = rng(H * 7 + S)
    q = r.standard_normal(H * D).astype(np.float32)
    Kc = r.standard_normal((C, Hkv * D)).astype(np.float32)
    Vc = r.standard_normal((C, Hkv * D)).astype(np.float32)
    if kv16:
        Kc, Vc = Kc.astype(np.float16), Vc.astype(np.float16)
    dt = np.float16 if kv16 else np.float32
    pq, po = ShaderProperty(mgr, H * D), ShaderProperty(mgr, H * D)
    pk, pv = ShaderProperty(mgr, C * Hkv * D, dt), ShaderProperty(mgr, C * Hkv * D, dt)
    pq.SetValue(q); pk.SetValue(Kc); pv.SetValue(Vc)
    call("nfai_hip_attn_decode", mgr.handle, pq.handle, pk.handle, pv.handle, po.handle, H, Hkv, D, S, C,
         _lib.F16 if kv16 else _lib.F32)
    K32, V32 = Kc.astype(np.float32), Vc.astype(np.float32)
    ref = orc.attn_wsum(orc.attn_softmax(orc.attn_scores(q, K32, H, Hkv, D, S)), V32, H, Hkv, D, S)
    np.testing.assert_allclose(po.GetValue(), ref, rtol=0, atol=3e-5)
    np.testing.assert_allclose(po.GetValue(), npo.attention(q, K32, V32, H, Hkv, D, S), rtol=0, atol=3e-5)


def test_attn_decode_online_softmax_spike(mgr):
    """Slice merge under a forced max jump: one key dominates inside a late slice (guide rule 26)."""
    from nfai_amd import _lib
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    H, Hkv, D, S, C = 24, 8, 128, 600, 640
    r = rng(4)
    q = r.standard_normal(H * D).astype(np.float32)
    Kc = (0.1 * r.standard_normal((C, Hkv * D))).astype(np.float32)
    Vc = r.standard_normal((C, Hkv * D)).astype(np.float32)
    Kc[555, :D] = 3.0 * q[:D]  # head 0's key at t=555 aligned with q -> score >> others
    pq, po = ShaderProperty(mgr, H * D), ShaderProperty(mgr, H * D)
    pk, pv = ShaderProperty(mgr, C * Hkv * D), ShaderProperty(mgr, C * Hkv * D)
    pq.SetValue(q); pk.SetValue(Kc); pv.SetValue(Vc)
    call("nfai_hip_attn_decode", mgr.handle, pq.handle, pk.handle, pv.handle, po.handle, H, Hkv, D, S, C, _lib.F32)
    ref = orc.attn_wsum(orc.attn_softmax(orc.attn_scores(q, Kc, H, Hkv, D, S)), Vc, H, Hkv, D, S)
    np.testing.assert_allclose(po.GetValue(), ref, rtol=0, atol=3e-5)
    np.testing.assert_allclose(po.GetValue()[:D], Vc[555, :D], rtol=0, atol=1e-3)  # head 0 ~ one-hot on t=555


def test_scratch_users_do_not_share_ranges(mgr):
    """the op-level entry points keep ticket words in the context scratch; lm_head + ArgMax (large winning row indices
    at V = 128,256), the ticket form of the fused attention (8B head shape, 32 slices at S >= 1024), the plain ArgMax and the
    top-k launch interleaved on ONE context must each stay exact — their workspaces are disjoint ranges now."""
    import ctypes as Cc
    from nfai_amd import _lib
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    r = rng(2024)
    V, E = 128256, 256
    base = (0.02 * r.standard_normal((4096, E))).astype(np.float16)
    W = np.tile(base, ((V + 4095) // 4096, 1))[:V].copy()
    x = r.standard_normal(E).astype(np.float32)
    W[V - 5] = (np.sign(x) * 0.05).astype(np.float16)          # the winner sits in the last workgroups: large partial indices
    tab = mgr.UploadWeight(1, W, V, E)
    px, pl, pi = ShaderProperty(mgr, E), ShaderProperty(mgr, V), ShaderProperty(mgr, 1, np.uint32)
    px.SetValue(x)
    H, Hkv, D, S, C = 32, 8, 128, 1500, 2048
    q = r.standard_normal(H * D).astype(np.float32)
    Kc = r.standard_normal((C, Hkv * D)).astype(np.float32)
    Vc = r.standard_normal((C, Hkv * D)).astype(np.float32)
    pq, po = ShaderProperty(mgr, H * D), ShaderProperty(mgr, H * D)
    pk, pv = ShaderProperty(mgr, C * Hkv * D), ShaderProperty(mgr, C * Hkv * D)
    pq.SetValue(q); pk.SetValue(Kc); pv.SetValue(Vc)
    att_ref = orc.attn_wsum(orc.attn_softmax(orc.attn_scores(q, Kc, H, Hkv, D, S)), Vc, H, Hkv, D, S)
    k = 40
    ids, probs = np.empty(k, np.uint32), np.empty(k, np.float32)
    for rep in range(3):
        call("nfai_hip_lmhead_argmax", mgr.handle, tab.handle, 1, px.handle, 0, 1e-5, pl.handle, pi.handle, V, E)
        lg = pl.GetValue()
        assert int(pi.GetValue()[0]) == V - 5 == int(np.argmax(lg))
        po.SetValue(np.zeros(H * D, np.float32))
        call("nfai_hip_attn_decode", mgr.handle, pq.handle, pk.handle, pv.handle, po.handle, H, Hkv, D, S, C, _lib.F32)
        np.testing.assert_allclose(po.GetValue(), att_ref, rtol=0, atol=3e-5)
        pi.SetValue(np.zeros(1, np.uint32))
        call("nfai_hip_lmhead_argmax", mgr.handle, tab.handle, 1, px.handle, 0, 1e-5, pl.handle, pi.handle, V, E)
        assert int(pi.GetValue()[0]) == V - 5
        call("nfai_hip_topk", mgr.handle, pl.handle, V, 0.5, k, ids.ctypes.data_as(Cc.POINTER(Cc.c_uint32)), probs.ctypes.data_as(Cc.POINTER(Cc.c_float)))
        _, ids_ref, probs_ref, _ = orc.topp(lg, 0.5, 0.95, k, 0.0)
        np.testing.assert_array_equal(ids, ids_ref)
        np.testing.assert_allclose(probs, probs_ref, rtol=1e-6)
        call("nfai_hip_argmax", mgr.handle, pl.handle, V, pi.handle)
        assert int(pi.GetValue()[0]) == V - 5
        po.SetValue(np.zeros(H * D, np.float32))
        call("nfai_hip_attn_decode", mgr.handle, pq.handle, pk.handle, pv.handle, po.handle, H, Hkv, D, S, C, _lib.F32)
        np.testing.assert_allclose(po.GetValue(), att_ref, rtol=0, atol=3e-5)


@pytest.mark.parametrize("E,N", [(2048, 2048), (3072, 3072), (4096, 1024), (256, 96)])
def test_gemv_fused_norm_residual(mgr, E, N):
    from nfai_amd import _lib
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    r = rng(E + N)
    W = (0.02 * r.standard_normal((N, E))).astype(np.float16)
    x = r.standard_normal(E).astype(np.float32)
    g = (1 + 0.1 * r.standard_normal(E)).astype(np.float32)
    res = r.standard_normal(N).astype(np.float32)
    pw = mgr.UploadWeight(_lib.F16, W, N, E)
    px, pg, pr, py = ShaderProperty(mgr, E), ShaderProperty(mgr, E), ShaderProperty(mgr, N), ShaderProperty(mgr, N)
    px.SetValue(x); pg.SetValue(g); pr.SetValue(res)
    xn = orc.rmsnorm(x, g, 1e-5)
    for gamma, resid, want in ((pg, pr, orc.add(res, orc.gemv_f16w(W, xn))), (pg, None, orc.gemv_f16w(W, xn)),
                               (None, pr, orc.add(res, orc.gemv_f16w(W, x)))):
        call("nfai_hip_gemv_fused", mgr.handle, pw.handle, _lib.F16, px.handle, gamma.handle if gamma else 0, 1e-5,
             resid.handle if resid else 0, py.handle, N, E)
        assert (np.abs(py.GetValue() - want) <= gemv_tol(W, xn if gamma else x) + 1e-6).all()


@pytest.mark.parametrize("E,F", [(2048, 8192), (3072, 8192), (4096, 14336), (256, 520)])
def test_gemv_gateup_silu(mgr, E, F):
    from nfai_amd import _lib
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    r = rng(E + F)
    Wg = (0.02 * r.standard_normal((F, E))).astype(np.float16)
    Wu = (0.02 * r.standard_normal((F, E))).astype(np.float16)
    x = r.standard_normal(E).astype(np.float32)
    g = (1 + 0.1 * r.standard_normal(E)).astype(np.float32)
    pg_, pu = mgr.UploadWeight(_lib.F16, Wg, F, E), mgr.UploadWeight(_lib.F16, Wu, F, E)
    px, pg, py = ShaderProperty(mgr, E), ShaderProperty(mgr, E), ShaderProperty(mgr, F)
    px.SetValue(x); pg.SetValue(g)
    call("nfai_hip_gemv_gateup_silu", mgr.handle, pg_.handle, pu.handle, _lib.F16, px.handle, pg.handle, 1e-5, py.handle, F, E)
    xn = orc.rmsnorm(x, g, 1e-5)
    want = orc.mul(orc.gemv_f16w(Wu, xn), orc.silu(orc.gemv_f16w(Wg, xn)))
    np.testing.assert_allclose(py.GetValue(), want, rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize("E,H,Hkv,D,C,pos,nfreq", [(2048, 32, 8, 64, 64, 0, None), (2048, 32, 8, 64, 64, 63, None),
                                                   (3072, 24, 8, 128, 640, 511, None), (3072, 24, 8, 128, 640, 100, 32),
                                                   (4096, 32, 8, 128, 32, 7, None), (256, 4, 2, 64, 16, 3, None)])
@pytest.mark.parametrize("kv16", [False, True])
def test_gemv_qkv_rope_kvwrite(mgr, E, H, Hkv, D, C, pos, nfreq, kv16):
    from nfai_amd import _lib
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    r = rng(E + pos)
    Wq = (0.02 * r.standard_normal((H * D, E))).astype(np.float16)
    Wk = (0.02 * r.standard_normal((Hkv * D, E))).astype(np.float16)
    Wv = (0.02 * r.standard_normal((Hkv * D, E))).astype(np.float16)
    x = r.standard_normal(E).astype(np.float32)
    g = (1 + 0.1 * r.standard_normal(E)).astype(np.float32)
    freqs = orc.rope_freqs(D, 500000.0, nfreq)
    dt = np.float16 if kv16 else np.float32
    pwq, pwk, pwv = (mgr.UploadWeight(_lib.F16, w, w.shape[0], E) for w in (Wq, Wk, Wv))
    px, pg, pf, pq = ShaderProperty(mgr, E), ShaderProperty(mgr, E), ShaderProperty(mgr, D // 2), ShaderProperty(mgr, H * D)
    pk, pv = ShaderProperty(mgr, C * Hkv * D, dt), ShaderProperty(mgr, C * Hkv * D, dt)
    sentinel = np.full(C * Hkv * D, 7.0, dt)
    px.SetValue(x); pg.SetValue(g); pf.SetValue(freqs); pk.SetValue(sentinel); pv.SetValue(sentinel)
    call("nfai_hip_gemv_qkv_rope", mgr.handle, pwq.handle, pwk.handle, pwv.handle, _lib.F16, px.handle, pg.handle, 1e-5,
         pf.handle, D, pq.handle, pk.handle, pv.handle, H, Hkv, D, pos, _lib.F16 if kv16 else _lib.F32, E)
    xn = orc.rmsnorm(x, g, 1e-5)
    q_ref = orc.rope(orc.gemv_f16w(Wq, xn), freqs, D, H, D, pos)
    k_ref = orc.rope(orc.gemv_f16w(Wk, xn), freqs, D, Hkv, D, pos)
    v_ref = orc.gemv_f16w(Wv, xn)
    tol = 2e-5 + 2e-6 * pos / 8
    np.testing.assert_allclose(pq.GetValue(), q_ref, rtol=0, atol=tol)
    Kg = pk.GetValue().reshape(C, Hkv * D).astype(np.float32)
    Vg = pv.GetValue().reshape(C, Hkv * D).astype(np.float32)
    kv_tol = tol + (2e-3 if kv16 else 0)
    np.testing.assert_allclose(Kg[pos], k_ref, rtol=0, atol=kv_tol)
    np.testing.assert_allclose(Vg[pos], v_ref, rtol=0, atol=kv_tol)
    assert (np.delete(Kg, pos, 0) == 7.0).all() and (np.delete(Vg, pos, 0) == 7.0).all()  # only row `pos` written


def test_error_paths(mgr):
    from nfai_amd import _lib
    from nfai_amd._lib import NfaiHipError, call
    from nfai_amd.hip import ShaderProperty
    a, b = ShaderProperty(mgr, 16), ShaderProperty(mgr, 8)
    with pytest.raises(NfaiHipError) as e:
        call("nfai_hip_add", mgr.handle, a.handle, b.handle, a.handle, 16)  # b too small
    assert e.value.code == _lib.ERR_INVALID and "needs" in str(e.value)
    with pytest.raises(NfaiHipError):
        call("nfai_hip_silu", mgr.handle, 12345, a.handle, 16)  # bogus handle
    with pytest.raises(NfaiHipError) as e:
        mgr.UploadWeight(_lib.Q4_K, np.zeros(100, np.uint8), 1, 100)  # K-quants need K % 256 == 0
    assert e.value.code == _lib.ERR_UNSUPPORTED


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 21, 22, 23, 24, 25, 26, 35, 38, 39, 42, 43, 44],
                         ids=["auto", "128x64", "128x128", "glds2", "glds3", "glds-n2", "glds-n3", "glds-n4", "glds-n3-pipe", "glds-n3-bk128", "glds-n2-bk128", "glds2-pipe",
                              "roles-n-b6", "roles-n-b9", "roles-w-b4", "roles-w-b6",
                              # round 3: eight (sixteen) waves per workgroup — 256 x 128 tiles, wave groups splitting the k-steps (ks2), BK 128
                              "256x128-w4-glds2", "256x128-w4-glds3", "256x128-w8-glds2", "256x128-w8-glds3", "128x128-ks2", "128x64-ks2", "256x128-w16-ks2",
                              "128x128-ks2-glds4", "128x64-bk128-ks2", "128x64-ks2-glds4", "256x128-w8-glds3-pipe", "256x128-w8-glds2-pipe"])
@pytest.mark.parametrize("M,N,K,res", [(512, 1024, 512, True), (200, 256, 384, False), (128, 384, 3072, True)])
def test_gemm_f16_variants(mgr, variant, M, N, K, res):
    """The prefill GEMM (MatrixMultiplyShader with inputRowCount = M, which the reference never exercises) in every tile /
    staging configuration against fp64 NumPy on the same fp16 operands: fp32 accumulation, so the error is summation-order
    noise: |d| <= 2e-6 * sqrt(K) * |A||W| scale + 1e-6.  Covers the direct-to-LDS kernels, ragged M and the residual."""
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    r = np.random.Generator(np.random.PCG64(100 * variant + M))
    A = r.standard_normal((M, K)).astype(np.float16)
    W = (0.05 * r.standard_normal((N, K))).astype(np.float16)
    R = r.standard_normal((M, N)).astype(np.float32) if res else None
    pa, pw = ShaderProperty(mgr, M * K, np.float16), ShaderProperty(mgr, N * K, np.float16)
    pc = ShaderProperty(mgr, M * N, np.float32)
    pa.SetValue(A.ravel()); pw.SetValue(W.ravel())
    pr = None
    if res:
        pr = ShaderProperty(mgr, M * N, np.float32)
        pr.SetValue(R.ravel())
    call("nfai_hip_gemm_f16", mgr.handle, pa.handle, pw.handle, pr.handle if res else 0, pc.handle, M, N, K, variant)
    got = pc.GetValue().reshape(M, N)
    want = A.astype(np.float64) @ W.astype(np.float64).T + (R.astype(np.float64) if res else 0.0)
    scale = float(np.abs(A.astype(np.float64)).mean() * np.abs(W.astype(np.float64)).mean() * K)
    assert np.abs(got - want).max() <= 2e-6 * np.sqrt(K) * scale + 1e-5, np.abs(got - want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [12, 13, 14, 15, 20, 27, 28, 29, 30, 31, 32, 33, 34, 36, 37, 40, 41, 45, 46, 47, 48],
                         ids=["128x80-glds3", "128x48-glds3", "128x80-glds4", "128x48-glds4", "128x80-roles-b6",
                              # round 3: two (four) wave groups on the tile, BK 64 / 128, deeper rings, pipelined reads, 64 x 96 tiles
                              "128x80-ks2", "128x48-ks2", "128x48-ks2-glds4", "128x96-2x2-ks2", "128x48-ks2-glds5", "128x48-bk128-ks2", "128x48-bk128-ks4",
                              "128x80-ks2-glds4", "128x48-ks2-glds6", "128x80-ks2-glds5", "128x80-bk128-ks2", "128x96-ks2-glds4", "128x48-bk128-ks2-pipe",
                              "128x80-bk128-ks2-pipe", "64x96-bk128-ks2", "64x96-ks2-glds4"])
@pytest.mark.parametrize("M,N,K,res", [(512, 960, 512, True), (200, 960, 384, False), (512, 5760, 3072, True)])
def test_gemm_f16_odd_tile_widths(mgr, variant, M, N, K, res):
    """Tile widths 80 and 48 (exactly 256 workgroups on the 5120- and 3072-column projections at 512 rows): the B tile's
    LDS-DMA instructions do not divide evenly over the four waves, so the waves wait on different counts."""
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    r = np.random.Generator(np.random.PCG64(100 * variant + M))
    A = r.standard_normal((M, K)).astype(np.float16)
    W = (0.05 * r.standard_normal((N, K))).astype(np.float16)
    R = r.standard_normal((M, N)).astype(np.float32) if res else None
    pa, pw = ShaderProperty(mgr, M * K, np.float16), ShaderProperty(mgr, N * K, np.float16)
    pc = ShaderProperty(mgr, M * N, np.float32)
    pa.SetValue(A.ravel()); pw.SetValue(W.ravel())
    pr = None
    if res:
        pr = ShaderProperty(mgr, M * N, np.float32)
        pr.SetValue(R.ravel())
    call("nfai_hip_gemm_f16", mgr.handle, pa.handle, pw.handle, pr.handle if res else 0, pc.handle, M, N, K, variant)
    got = pc.GetValue().reshape(M, N)
    want = A.astype(np.float64) @ W.astype(np.float64).T + (R.astype(np.float64) if res else 0.0)
    scale = float(np.abs(A.astype(np.float64)).mean() * np.abs(W.astype(np.float64)).mean() * K)
    assert np.abs(got - want).max() <= 2e-6 * np.sqrt(K) * scale + 1e-5, np.abs(got - want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("H,Hkv,D,T,pos0", [(24, 8, 128, 512, 0), (32, 8, 64, 200, 0), (4, 2, 128, 37, 64), (32, 8, 128, 130, 126), (2, 2, 64, 16, 0)],
                         ids=["3b-512", "1b-200-ragged", "tiny-chunk2", "8b-chunk-at-126", "one-tile"])
def test_attn_prefill_one_launch(mgr, H, Hkv, D, T, pos0):
    """Causal attention of a prompt chunk in one launch (k_attn_prefill: K.Q^T and V^T.P^T on the matrix cores, probabilities
    kept in registers, online softmax) against fp64 NumPy on the same fp16 operands: query t sees keys 0..pos0+t, softmax of
    q.k/sqrt(D) (AttentionScoreCalculationShader.cs:93, AttentionSoftmaxShader.cs:148-176), weighted sum of V.  The
    probabilities are rounded to fp16 before they multiply V (as in the GEMM path): |d| <= 2e-3 * max|V| + fp16 output rounding."""
    from nfai_amd._lib import NfaiHipError, call
    from nfai_amd.hip import ShaderProperty
    r = rng(900 + T + pos0)
    S = pos0 + T
    Spad = (S + 63) // 64 * 64
    G = H // Hkv
    Q = r.standard_normal((T, H, D)).astype(np.float16)
    K = np.zeros((Hkv, Spad, D), np.float16)
    V = np.zeros((Hkv, Spad, D), np.float16)
    K[:, :S] = r.standard_normal((Hkv, S, D)).astype(np.float16)
    V[:, :S] = r.standard_normal((Hkv, S, D)).astype(np.float16)
    Vt = np.ascontiguousarray(V.transpose(0, 2, 1))
    pq, pk = ShaderProperty(mgr, Q.size, np.float16), ShaderProperty(mgr, K.size, np.float16)
    pv, po = ShaderProperty(mgr, Vt.size, np.float16), ShaderProperty(mgr, Q.size, np.float16)
    pq.SetValue(Q.ravel()); pk.SetValue(K.ravel()); pv.SetValue(Vt.ravel())
    call("nfai_hip_attn_prefill", mgr.handle, pq.handle, pk.handle, pv.handle, po.handle, T, H, Hkv, D, Spad, pos0)
    got = po.GetValue().reshape(T, H, D).astype(np.float64)
    want = np.zeros((T, H, D))
    q64, k64, v64 = Q.astype(np.float64), K.astype(np.float64), V.astype(np.float64)
    for h in range(H):
        sc = q64[:, h] @ k64[h // G, :S].T / np.sqrt(D)          # [T][S]
        mask = np.arange(S)[None, :] <= (pos0 + np.arange(T))[:, None]
        sc = np.where(mask, sc, -np.inf)
        p = np.exp(sc - sc.max(axis=1, keepdims=True))
        p /= p.sum(axis=1, keepdims=True)
        want[:, h] = p @ v64[h // G, :S]
    tol = 2e-3 * float(np.abs(v64).max()) + 2e-3 * np.abs(want).max()
    assert np.abs(got - want).max() <= tol, (np.abs(got - want).max(), tol)
    with pytest.raises(NfaiHipError, match="bad shape"):
        call("nfai_hip_attn_prefill", mgr.handle, pq.handle, pk.handle, pv.handle, po.handle, T, H, Hkv, D, Spad, Spad)  # pos0 + T > Spad


def _silu64(x):
    return x / (1.0 + np.exp(-x))


WIDE = [0, 2, 3, 4, 11, 18, 19, 24, 25, 43]     # 128 x 128 / 256 x 128 tile configurations (what gemm_pick takes for wide N at prefill sizes)
NARROW = [1, 5, 6, 7, 8, 16, 17, 26, 39, 42]    # 128 x 64 (the last three: two wave groups)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", WIDE + NARROW)
@pytest.mark.parametrize("M,N,K", [(512, 1024, 512), (200, 256, 384), (384, 2048, 3072)])
def test_gemm_f16_fp16_epilogue(mgr, variant, M, N, K):
    """The fp16 epilogue (P.V writes the Wo GEMM's A operand; the K-quant widening path) in every tile configuration,
    the 128 x 128 direct-to-LDS ones included: fp64 product rounded once to fp16 => half an fp16 ulp + summation noise."""
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    r = rng(300 + variant + M)
    A = r.standard_normal((M, K)).astype(np.float16)
    W = (0.05 * r.standard_normal((N, K))).astype(np.float16)
    pa, pw = ShaderProperty(mgr, M * K, np.float16), ShaderProperty(mgr, N * K, np.float16)
    pc = ShaderProperty(mgr, M * N, np.float16)
    pa.SetValue(A.ravel()); pw.SetValue(W.ravel())
    call("nfai_hip_gemm_f16_ex", mgr.handle, pa.handle, pw.handle, 0, 0, pc.handle, M, N, K, variant, 1, 1, 1, 0, 0)
    got = pc.GetValue().reshape(M, N).astype(np.float64)
    want = A.astype(np.float64) @ W.astype(np.float64).T
    assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max() + 1e-4, np.abs(got - want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", WIDE + NARROW)
@pytest.mark.parametrize("M,F,K", [(512, 1024, 512), (200, 128, 384), (256, 8192, 3072)])
def test_gemm_f16_silu_up_epilogue(mgr, variant, M, F, K):
    """gate | up GEMM with act = up * silu(gate) formed in the epilogue (SiLUShader.cs:121-123 + ElementWiseMultiplicationShader.cs:137)
    at Llama-3.2-3B's gate|up shape (F = 8192, K = 3072: the launch that dominates the 512-token prefill) and small / ragged ones."""
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    r = rng(400 + variant + M)
    A = r.standard_normal((M, K)).astype(np.float16)
    Wg = (0.05 * r.standard_normal((F, K))).astype(np.float16)
    Wu = (0.05 * r.standard_normal((F, K))).astype(np.float16)
    pa = ShaderProperty(mgr, M * K, np.float16)
    pg, pu = ShaderProperty(mgr, F * K, np.float16), ShaderProperty(mgr, F * K, np.float16)
    pc = ShaderProperty(mgr, M * F, np.float16)
    pa.SetValue(A.ravel()); pg.SetValue(Wg.ravel()); pu.SetValue(Wu.ravel())
    call("nfai_hip_gemm_f16_ex", mgr.handle, pa.handle, pg.handle, pu.handle, 0, pc.handle, M, 2 * F, K, variant, 2, 1, 1, 0, 0)
    got = pc.GetValue().reshape(M, F).astype(np.float64)
    g = A.astype(np.float64) @ Wg.astype(np.float64).T
    u = A.astype(np.float64) @ Wu.astype(np.float64).T
    want = u * _silu64(g)
    assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max() + 1e-4, np.abs(got - want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 6, 11])
@pytest.mark.parametrize("pos0", [0, 192])
def test_gemm_f16_causal_attention_pair(mgr, variant, pos0):
    """The two head-batched attention GEMMs of the prefill at Llama-3.2-3B's head geometry (H = 24, Hkv = 8, D = 128), T = 512:
    scores = Q.K^T with the tiles above the causal diagonal skipped (only s <= pos0 + t is ever read by the softmax, so only
    that part is compared), and att = P.V^T-form product with the K tiles past the last unmasked key skipped (P is zero there,
    as the causal softmax leaves it), fp16 output.  pos0 > 0 = a later chunk of a chunked prompt."""
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    H, Hkv, D, T = 24, 8, 128, 512
    G = H // Hkv
    S = pos0 + T
    Spad = (S + 63) // 64 * 64
    r = rng(500 + variant + pos0)
    Q = r.standard_normal((H, T, D)).astype(np.float16)
    Kh = r.standard_normal((Hkv, Spad, D)).astype(np.float16)
    pq, pk = ShaderProperty(mgr, Q.size, np.float16), ShaderProperty(mgr, Kh.size, np.float16)
    psc = ShaderProperty(mgr, H * T * Spad, np.float32)
    pq.SetValue(Q.ravel()); pk.SetValue(Kh.ravel())
    psc.SetValue(np.full(H * T * Spad, 7.0, np.float32))  # skipped tiles must stay untouched
    call("nfai_hip_gemm_f16_ex", mgr.handle, pq.handle, pk.handle, 0, 0, psc.handle, T, Spad, D, variant, 0, H, G, 1, pos0)
    sc = psc.GetValue().reshape(H, T, Spad)
    t_idx, s_idx = np.arange(T)[:, None], np.arange(Spad)[None, :]
    live = s_idx <= pos0 + t_idx
    for h in range(H):
        want = Q[h].astype(np.float64) @ Kh[h // G].astype(np.float64).T
        d = np.abs(sc[h] - want)[live].max()
        assert d <= 2e-6 * np.sqrt(D) * D + 1e-4, (h, d)
    # probabilities: causal softmax of the scores (zeros past the diagonal), then P.V with the masked K tiles skipped
    P = np.zeros((H, T, Spad), np.float16)
    for h in range(H):
        x = np.where(live, sc[h] / np.sqrt(D), -np.inf)
        e = np.exp(x - x.max(axis=1, keepdims=True))
        P[h] = (e / e.sum(axis=1, keepdims=True)).astype(np.float16)
    Vt = r.standard_normal((Hkv, D, Spad)).astype(np.float16)
    pp, pv = ShaderProperty(mgr, P.size, np.float16), ShaderProperty(mgr, Vt.size, np.float16)
    po = ShaderProperty(mgr, H * T * D, np.float16)
    pp.SetValue(P.ravel()); pv.SetValue(Vt.ravel())
    v2 = variant if D % 128 == 0 or variant in (0, 1, 5, 6, 7, 8) else 1
    call("nfai_hip_gemm_f16_ex", mgr.handle, pp.handle, pv.handle, 0, 0, po.handle, T, D, Spad, v2, 1, H, G, 2, pos0)
    att = po.GetValue().reshape(H, T, D).astype(np.float64)
    for h in range(H):
        want = P[h].astype(np.float64) @ Vt[h // G].astype(np.float64).T
        assert np.abs(att[h] - want).max() <= 2e-3 * np.abs(want).max() + 1e-4, (h, np.abs(att[h] - want).max())


def _rms(x, g, eps=1e-5):
    x = x.astype(np.float64)
    return (x / np.sqrt((x * x).mean() + eps)) * g.astype(np.float64)


@pytest.mark.gpu
@pytest.mark.parametrize("E,F,H,Hkv,D,with_qkv", [(512, 1024, 4, 2, 128, True), (512, 1024, 4, 2, 128, False), (2048, 8192, 32, 8, 64, True),
                                                  (3072, 8192, 24, 8, 128, True), (4096, 14336, 32, 8, 128, True)],
                         ids=["tiny", "tiny-no-qkv", "1b", "3b", "8b"])
def test_engine_block_op_level(mgr, E, F, H, Hkv, D, with_qkv):
    """One launch of the weight-streaming engine against fp64 NumPy, stage by stage (the hand-off vectors h and act are read back
    from the granules): Wo + residual, RMSNorm + gate|up + SiLU*up, Wdown + residual, RMSNorm + next q|k|v + RoPE + KV rows
    (TransformerBlock.cs:150-181, :129-141).  fp32 summation-order tolerances as for the GEMV family; two calls bit-equal."""
    from nfai_amd._lib import call
    from nfai_amd.hip import ShaderProperty
    r = rng(E + F + int(with_qkv))
    HD, KD = H * D, Hkv * D
    def mat(n, k, s=0.02):
        return (s * r.standard_normal((n, k))).astype(np.float16)
    Wo, Wg, Wu, Wd = mat(E, HD), mat(F, E), mat(F, E), mat(E, F)
    Wq, Wk, Wv = mat(HD, E), mat(KD, E), mat(KD, E)
    att = r.standard_normal(HD).astype(np.float32)
    x = r.standard_normal(E).astype(np.float32)
    g1 = (1 + 0.1 * r.standard_normal(E)).astype(np.float32)
    g2 = (1 + 0.1 * r.standard_normal(E)).astype(np.float32)
    pos = 5
    freqs = (1.0 / 500000.0 ** (np.arange(D // 2) / (D / 2))).astype(np.float32)
    def up(a, dt):
        p = ShaderProperty(mgr, a.size, dt)
        p.SetValue(a.ravel())
        return p
    pWo, pWg, pWu, pWd = (up(a, np.float16) for a in (Wo, Wg, Wu, Wd))
    pWq, pWk, pWv = (up(a, np.float16) for a in (Wq, Wk, Wv))
    patt, px, pg1, pg2, pfr = (up(a, np.float32) for a in (att, x, g1, g2, freqs))
    pxo, pq = ShaderProperty(mgr, E), ShaderProperty(mgr, HD)
    pkc, pvc = ShaderProperty(mgr, (pos + 1) * KD), ShaderProperty(mgr, (pos + 1) * KD)
    nwords = (2 * E + F) * 2 + 16
    psc = ShaderProperty(mgr, nwords, np.uint32)
    psc.SetValue(np.zeros(nwords, np.uint32))
    def run():
        q = (pWq.handle, pWk.handle, pWv.handle, pg2.handle, pfr.handle) if with_qkv else (0, 0, 0, 0, 0)
        o = (pq.handle, pkc.handle, pvc.handle) if with_qkv else (0, 0, 0)
        call("nfai_hip_engine_block", mgr.handle, pWo.handle, pWg.handle, pWu.handle, pWd.handle, patt.handle, px.handle, pg1.handle,
             1e-5, E, F, HD, *q, D, *o, H, Hkv, D, pos, 0, pxo.handle, psc.handle)
        return pxo.GetValue().copy(), psc.GetValue().copy()
    xo, sc = run()
    gran = sc[: (2 * E + F) * 2].reshape(-1, 2)
    vals = gran[:, 0].copy().view(np.float32)
    assert (gran[: E + F, 1] == 1).all()  # every granule of h and act carries the first call's epoch
    f64 = np.float64
    h = x.astype(f64) + Wo.astype(f64) @ att.astype(f64)
    xa = _rms(h, g1)
    gate, upv = Wg.astype(f64) @ xa, Wu.astype(f64) @ xa
    act = upv * _silu64(gate)
    want_x = h + Wd.astype(f64) @ act
    tol_h = gemv_tol(Wo, att).max() + 1e-6
    assert np.abs(vals[:E] - h).max() <= tol_h, np.abs(vals[:E] - h).max()
    assert np.abs(vals[E:E + F] - act).max() <= 2e-5 * max(1.0, np.abs(act).max()), np.abs(vals[E:E + F] - act).max()
    assert np.abs(xo - want_x).max() <= 5e-5 * max(1.0, np.abs(want_x).max()), np.abs(xo - want_x).max()
    if with_qkv:
        assert (gran[E + F:, 1] == 1).all()
        np.testing.assert_array_equal(vals[E + F:], xo)  # the granule copy and the plain copy of the block output
        xn = _rms(want_x, g2)
        qkv = np.concatenate([Wq, Wk, Wv]).astype(f64) @ xn
        th = freqs.astype(f64) * pos
        def rope(v, nh):
            v = v.reshape(nh, D // 2, 2).copy()
            a, b = v[..., 0].copy(), v[..., 1].copy()
            v[..., 0] = a * np.cos(th) - b * np.sin(th)
            v[..., 1] = a * np.sin(th) + b * np.cos(th)
            return v.reshape(-1)
        tol = 1e-4 * max(1.0, np.abs(qkv).max())
        assert np.abs(pq.GetValue() - rope(qkv[:HD], H)).max() <= tol
        assert np.abs(pkc.GetValue()[pos * KD:] - rope(qkv[HD:HD + KD], Hkv)).max() <= tol
        assert np.abs(pvc.GetValue()[pos * KD:] - qkv[HD + KD:]).max() <= tol
    xo2, sc2 = run()  # second call: epoch 2, same values bit for bit
    np.testing.assert_array_equal(xo, xo2)
    assert (sc2[: (E + F) * 2].reshape(-1, 2)[:, 1] == 2).all()
