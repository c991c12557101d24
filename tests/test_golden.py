"""Committed golden fixtures (tests/golden/*.npz, generator tests/golden/make_golden.py).

CPU: the oracle reproduces them bit for bit (drift guard; also on a different host/compiler, since
the oracle is built with -ffp-contract=off and no fast-math).  GPU: the HIP path meets the stated
tolerances against the committed numbers without needing the oracle at all.  The reference has no
vectors of its own for this path — see DESIGN.md §2 "parity unpinned"."""
import os

import numpy as np
import pytest

from nfai_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLD, name)))


def test_oracle_reproduces_op_fixtures():
    import oracle as orc
    g = load("ops.npz")
    H, Hkv, D, C, S = (int(v) for v in g["dims"])
    np.testing.assert_array_equal(orc.rmsnorm(g["x"], g["gamma"], 1e-5), g["rmsnorm"])
    np.testing.assert_array_equal(orc.gemv_f16w(g["W"], g["x"]), g["gemv"])
    np.testing.assert_array_equal(orc.gemv(g["W"].astype(np.float32), g["x"]), g["gemv"])  # widened fp32 == fp16 storage
    np.testing.assert_array_equal(orc.rope(g["q"], g["freqs"], D, H, D, 7), g["rope_q_pos7"])
    sc = orc.attn_scores(g["q"], g["Kc"], H, Hkv, D, S)
    np.testing.assert_array_equal(sc, g["scores"])
    np.testing.assert_array_equal(orc.attn_softmax(sc), g["softmax"])
    np.testing.assert_array_equal(orc.attn_wsum(g["softmax"], g["Vc"], H, Hkv, D, S), g["wsum"])
    np.testing.assert_array_equal(orc.silu(g["a"]), g["silu"])
    np.testing.assert_array_equal(orc.mul(g["a"], g["b"]), g["mul"])
    np.testing.assert_array_equal(orc.add(g["a"], g["b"]), g["add"])
    assert orc.argmax(g["a"]) == int(g["argmax"][0])
    np.testing.assert_array_equal(orc.quantize_q4k(g["kq_src"]), g["q4k_blocks"])
    np.testing.assert_array_equal(orc.quantize_q6k(g["kq_src"]), g["q6k_blocks"])
    np.testing.assert_array_equal(orc.dequant_q4k(g["q4k_blocks"], g["kq_src"].size), g["q4k_dequant"])
    np.testing.assert_array_equal(orc.dequant_q6k(g["q6k_blocks"], g["kq_src"].size), g["q6k_dequant"])


@pytest.mark.parametrize("name,dims", [("tiny_llama.npz", synth.TINY), ("tiny_llama_d128.npz", synth.TINY_D128)])
def test_oracle_reproduces_model_fixtures(name, dims):
    import oracle as orc
    g = load(name)
    w = synth.make_weights(dims, seed=int(g["seed"][0]), std=0.05)
    m = orc.OracleLlama(orc.LlamaDesc(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, C=int(g["C"][0])), w)
    seq = list(g["prompt"]) + list(g["greedy"])
    for i, t in enumerate(seq):
        np.testing.assert_array_equal(m.step(int(t)), g["logits"][i])
        if i >= len(g["prompt"]) - 1 and i < len(seq) - 1:
            assert orc.argmax(g["logits"][i]) == int(seq[i + 1])  # the stored continuation IS the greedy one
    np.testing.assert_array_equal(m.hidden(), g["hidden_last"])


@pytest.mark.gpu
@pytest.mark.parametrize("name,dims", [("tiny_llama.npz", synth.TINY), ("tiny_llama_d128.npz", synth.TINY_D128)])
@pytest.mark.parametrize("mode", ["graph", "unfused"])
def test_hip_path_against_model_fixtures(name, dims, mode):
    from nfai_amd.hip import HipBufferManager
    from nfai_amd.llama_model import LlamaModel
    g = load(name)
    w = synth.make_weights(dims, seed=int(g["seed"][0]), std=0.05)
    mgr = HipBufferManager(0)
    m = LlamaModel(mgr, synth.make_metadata(dims), w, int(g["C"][0]), unfused=mode == "unfused")
    prompt = [int(t) for t in g["prompt"]]
    lg = None
    for i, t in enumerate(prompt):
        lg, am = m.Step(t)
        want = g["logits"][i]
        assert np.abs(lg - want).max() <= 5e-4 * max(1.0, float(np.abs(want).max()))
    got = []
    tok = int(np.argmax(lg))
    for j in range(len(g["greedy"])):
        got.append(tok)
        lg, tok = m.Step(tok)
        want = g["logits"][len(prompt) + j]
        assert np.abs(lg - want).max() <= 5e-4 * max(1.0, float(np.abs(want).max()))
    assert got == [int(t) for t in g["greedy"]]  # identical greedy tokens
    np.testing.assert_allclose(m.Read(0, dims.E), g["hidden_last"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(m.ReadKV(0, False, len(prompt) + len(got) - 1), g["k_last_l0"], rtol=0, atol=1e-3)
    m.Dispose()
    mgr.Dispose()


@pytest.mark.gpu
def test_hip_ops_against_op_fixtures():
    from nfai_amd import _lib
    from nfai_amd._lib import call
    from nfai_amd.hip import HipBufferManager, ShaderProperty
    from nfai_amd.shaders import MatrixMultiplyShader, RMSNormShader, RoPEShader
    g = load("ops.npz")
    H, Hkv, D, C, S = (int(v) for v in g["dims"])
    mgr = HipBufferManager(0)
    n = RMSNormShader(mgr, g["x"].size, g["gamma"], 1e-5)
    n.GetInputProperty().SetValue(g["x"])
    n.Compute()
    np.testing.assert_allclose(n.GetOutputs(), g["rmsnorm"], rtol=3e-6, atol=1e-7)
    mm = MatrixMultiplyShader(mgr, 1, g["W"].shape[1], g["W"].shape[0], g["W"])
    mm.GetInputProperty().SetValue(g["x"])
    mm.Compute()
    np.testing.assert_allclose(mm.GetOutputs(), g["gemv"], rtol=0, atol=2e-5)
    rp = RoPEShader(mgr, H * D, H * D, g["freqs"], D, H)
    rp.GetInputProperty().SetValue(g["q"])
    rp.Compute(7)
    np.testing.assert_allclose(rp.GetOutputs(), g["rope_q_pos7"], rtol=0, atol=3e-6)
    pq, po = ShaderProperty(mgr, H * D), ShaderProperty(mgr, H * D)
    pk, pv = ShaderProperty(mgr, C * Hkv * D), ShaderProperty(mgr, C * Hkv * D)
    pq.SetValue(g["q"]); pk.SetValue(g["Kc"]); pv.SetValue(g["Vc"])
    call("nfai_hip_attn_decode", mgr.handle, pq.handle, pk.handle, pv.handle, po.handle, H, Hkv, D, S, C, _lib.F32)
    np.testing.assert_allclose(po.GetValue(), g["wsum"], rtol=0, atol=3e-5)
    for qt, blocks, deq in ((12, g["q4k_blocks"], g["q4k_dequant"]), (14, g["q6k_blocks"], g["q6k_dequant"])):
        rows, cols = g["kq_src"].shape
        x = np.linspace(-1, 1, cols).astype(np.float32)
        op = MatrixMultiplyShader(mgr, 1, cols, rows, None)
        op.GetWeightProperty().set(blocks, qt, rows, cols)
        op.GetInputProperty().SetValue(x)
        op.Compute()
        np.testing.assert_allclose(op.GetOutputs(), deq.reshape(rows, cols) @ x, rtol=0, atol=2e-5)
    mgr.Dispose()
