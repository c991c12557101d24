"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
include/nfai_hip.h declares; the ctypes binding covers the same set; without a GPU the product path
fails loudly (no CPU fallback); host-side helpers behave.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "nfai_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nfai_hip_\w+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from nfai_amd import build as hb, _lib
    hb.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    names = header_functions()
    assert len(names) >= 50
    raw = ctypes.CDLL(os.path.join(ROOT, "nfai_amd", "csrc", "libnfai_hip.so"))
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing
    assert lib.nfai_hip_abi_version() == 1


def test_binding_covers_the_header():
    from nfai_amd import _lib
    bound = set(_lib.SIGNATURES) | {"nfai_hip_last_error", "nfai_hip_abi_version"}
    assert bound == set(header_functions())


def test_every_entry_point_cites_the_reference():
    src = open(os.path.join(ROOT, "include", "nfai_hip.h")).read()
    assert len(re.findall(r"\.cs:\d+", src)) >= 30


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="needs a box WITHOUT a GPU")
def test_no_gpu_is_a_loud_error_not_a_fallback(lib):
    from nfai_amd._lib import NfaiHipError
    from nfai_amd.hip import HipBufferManager
    with pytest.raises(NfaiHipError, match="no CPU fallback"):
        HipBufferManager(0)


def test_invalid_handles_are_errors_not_crashes(lib):
    from nfai_amd import _lib
    for name, args in (("nfai_hip_ctx_synchronize", (12345,)), ("nfai_hip_buf_free", (0, 0)),
                       ("nfai_hip_llama_reset", (987654321,)), ("nfai_hip_ctx_destroy", (0,))):
        with pytest.raises(_lib.NfaiHipError, match="invalid"):
            _lib.call(name, *args)
    n = ctypes.c_uint64()
    _lib.call("nfai_hip_weight_bytes", _lib.Q4_K, 10, 512, ctypes.byref(n))
    assert n.value == 10 * 2 * 144
    _lib.call("nfai_hip_weight_bytes", _lib.Q6_K, 3, 256, ctypes.byref(n))
    assert n.value == 3 * 210
    with pytest.raises(_lib.NfaiHipError):
        _lib.call("nfai_hip_weight_bytes", _lib.Q4_K, 1, 100, ctypes.byref(n))
    with pytest.raises(_lib.NfaiHipError):
        _lib.call("nfai_hip_weight_bytes", 7, 1, 256, ctypes.byref(n))  # Q5_1: named by Parser.cs:262-293, no kernel


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under nfai_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "nfai_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f
                assert not re.search(r"#\s*include\s*[\"<][^\">]*oracle", txt), f       # no source inclusion
                assert "libnfai_oracle" not in txt and not re.search(r"\borc_\w+\s*\(", txt), f  # no linkage / calls


def test_sampling_utils():
    from nfai_amd.llama_model import SamplingUtils
    v = np.array([0.1, 3.0, 3.0, -1.0], np.float32)
    assert SamplingUtils.ArgMax(v) == 1
    np.testing.assert_allclose(SamplingUtils.Softmax(v).sum(), 1.0, rtol=1e-6)
    rng = np.random.default_rng(0)
    picks = {SamplingUtils.TopP(np.array([10.0, 9.9, -50.0, -50.0], np.float32), rng=rng) for _ in range(200)}
    assert picks <= {0, 1} and len(picks) == 2  # nucleus keeps the two likely tokens only
    assert SamplingUtils.TopP(np.array([50.0, 0.0, 0.0], np.float32), rng=rng) == 0


def test_synth_shapes_match_published_dims():
    from nfai_amd import synth
    assert synth.LLAMA_32_1B.n_params_read_per_token() == 1_235_746_816 - 0
    assert synth.LLAMA_32_3B.n_params_read_per_token() * 2 == 6_425_149_440 - 0
    md = synth.make_metadata(synth.LLAMA_32_3B)
    from nfai_amd.llama_model import dims_from_metadata
    shapes = {k: np.empty(v, np.float16) if len(v) == 1 else np.lib.stride_tricks.as_strided(np.zeros(1, np.float16), v, (0, 0))
              for k, v in synth.LLAMA_32_3B.shapes().items() if k.startswith(("token_embd", "blk.0."))}
    d = dims_from_metadata(md, shapes)
    assert (d["E"], d["L"], d["H"], d["Hkv"], d["D"], d["F"], d["V"]) == (3072, 28, 24, 8, 128, 8192, 128256)


# ---- the C# binding (csharp/NFAI.HIP): cannot be compiled here (no .NET SDK), so its consistency with the C ABI is checked as text ----
def _split_args(s):
    """top-level comma split of a C# / C argument list"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([<{":
            depth += 1
        elif ch in ")]>}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _csharp_decls():
    src = open(os.path.join(ROOT, "csharp", "NFAI.HIP", "NativeMethods.g.cs")).read()
    return {m.group(1): _split_args(m.group(2)) for m in re.finditer(r"internal static partial \w+ (nfai_hip_\w+)\((.*?)\);", src)}


def test_csharp_declarations_match_the_header():
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_csharp_bindings as gen
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_csharp_bindings.py"), "--check"]).returncode == 0
    decls = _csharp_decls()
    header = {name: params for name, _, params in gen.parse_header()}
    assert set(decls) == set(header) == set(header_functions())
    for name, params in header.items():
        assert len(decls[name]) == len(params), name


def test_csharp_sources_call_declared_entry_points_and_have_no_stubs():
    decls = _csharp_decls()
    cs_dir = os.path.join(ROOT, "csharp", "NFAI.HIP")
    files = [f for f in os.listdir(cs_dir) if f.endswith(".cs") and f != "NativeMethods.g.cs"]
    assert {"HipBufferManager.cs", "HipShaderProperty.cs", "HipShaders.cs", "HipLlamaModelFactory.cs", "HipPipeline.cs", "Native.cs"} <= set(files)
    called = set()
    for f in files:
        src = open(os.path.join(cs_dir, f)).read()
        code = re.sub(r"//[^\n]*", "", src)
        assert "NotImplementedException" not in code, f  # only the reference throws it (LlamaModel.cs:70-74)
        for m in re.finditer(r"Native\.(nfai_hip_\w+)\(", code):
            name, i, depth = m.group(1), m.end(), 1
            while depth:  # matching parenthesis of the call
                depth += {"(": 1, ")": -1}.get(code[i], 0)
                i += 1
            args = _split_args(code[m.end():i - 1])
            assert name in decls, (f, name)
            assert len(args) == len(decls[name]), (f, name, args)
            called.add(name)
    # the operator surface (ten op classes + block), buffers, model, pipeline are all bound to something real
    for need in ("nfai_hip_embed", "nfai_hip_rmsnorm", "nfai_hip_gemv", "nfai_hip_rope", "nfai_hip_attn_scores", "nfai_hip_attn_softmax",
                 "nfai_hip_attn_wsum", "nfai_hip_silu", "nfai_hip_mul", "nfai_hip_add", "nfai_hip_buf_alloc", "nfai_hip_buf_upload",
                 "nfai_hip_buf_download", "nfai_hip_buf_copy", "nfai_hip_weight_upload", "nfai_hip_llama_create", "nfai_hip_llama_set_tensor",
                 "nfai_hip_llama_finalize", "nfai_hip_llama_decode_step", "nfai_hip_pp_init", "nfai_hip_pp_send_hidden"):
        assert need in called, need
    shaders = open(os.path.join(cs_dir, "HipShaders.cs")).read()
    for cls in ("HipTokenEmbedShader", "HipRMSNormShader", "HipMatrixMultiplyShader", "HipRoPEShader", "HipAttentionScoreCalculationShader",
                "HipAttentionSoftmaxShader", "HipAttentionWeightedValueSumShader", "HipSiLUShader", "HipElementWiseMultiplicationShader",
                "HipTransformerBlock"):
        assert f"class {cls}" in shaders, cls
    prop = open(os.path.join(cs_dir, "HipShaderProperty.cs")).read()
    for member in ("BindShaderProprty", "SetValue", "GetValue", "TransferTo", "Count"):  # ShaderProperty.cs:20-182
        assert member in prop, member


def _build_c_driver(tmp_path):
    import subprocess
    from nfai_amd import build as hb
    from oracle import c_oracle
    c_oracle.build()  # oracle/libnfai_oracle.so (building the checker is not using it)
    hb.build()
    assert os.path.exists(os.path.join(ROOT, "oracle", "libnfai_oracle.so"))
    exe = str(tmp_path / "run_llama")
    r = subprocess.run(["gcc", "-O2", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "c_driver", "run_llama.c"), "-o", exe,
                        "-L", os.path.join(ROOT, "nfai_amd", "csrc"), "-lnfai_hip", "-L", os.path.join(ROOT, "oracle"), "-lnfai_oracle", "-lm"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_driver_compiles_against_the_header(tmp_path):
    """A plain C host (tests/c_driver/run_llama.c: LlamaModelFactory.TryCreate + LlamaModel.RunAsync over the C ABI) builds with gcc
    against include/nfai_hip.h and links libnfai_hip.so: the header is valid C and every symbol it uses is exported."""
    _build_c_driver(tmp_path)


@pytest.mark.gpu
def test_c_driver_runs_a_model_through_the_c_abi(tmp_path):
    """The C host on the GPU: 48 tokens of a small Llama-shaped model, logits against the oracle at every step, identical greedy
    tokens, the device-side greedy loop, the KV-capacity error.  No Python, torch or HIP header in that process."""
    import subprocess
    exe = _build_c_driver(tmp_path)
    env = dict(os.environ, LD_LIBRARY_PATH=os.pathsep.join([os.path.join(ROOT, "nfai_amd", "csrc"), os.path.join(ROOT, "oracle"),
                                                             os.environ.get("LD_LIBRARY_PATH", "")]))
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("ok"), r.stdout


@pytest.mark.gpu
def test_canary_padded_debug_buffers_in_a_child_process():
    """NFAI_HIP_DEBUG_CANARY=1 (read once per process): buffers sit between two guards.  A clean run of real kernels (GEMV with a
    ragged row count, attention, argmax, top-k) leaves them intact; a write past the end of a buffer — made here through a wrapped
    alias that is 64 bytes too long — is reported by the next synchronize and by the free, naming the buffer."""
    import subprocess
    import sys
    code = r"""
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r)
from nfai_amd import _lib
from nfai_amd._lib import call, NfaiHipError
from nfai_amd.hip import HipBufferManager, ShaderProperty
from nfai_amd.shaders import MatrixMultiplyShader
mgr = HipBufferManager(0)
r = np.random.default_rng(3)
W = (0.02 * r.standard_normal((130, 1000))).astype(np.float16)
op = MatrixMultiplyShader(mgr, 1, 1000, 130, W)
op.GetInputProperty().SetValue(r.standard_normal(1000).astype(np.float32))
op.Compute()
y = op.GetOutputs()
v = r.standard_normal(5000).astype(np.float32)
pv, pi = ShaderProperty(mgr, v.size), ShaderProperty(mgr, 1, np.uint32)
pv.SetValue(v)
call("nfai_hip_argmax", mgr.handle, pv.handle, v.size, pi.handle)
ids, probs = np.empty(40, np.uint32), np.empty(40, np.float32)
call("nfai_hip_topk", mgr.handle, pv.handle, v.size, 0.5, 40, ids.ctypes.data_as(C.POINTER(C.c_uint32)), probs.ctypes.data_as(C.POINTER(C.c_float)))
call("nfai_hip_ctx_synchronize", mgr.handle)            # every guard intact after real kernels
assert int(pi.GetValue()[0]) == int(np.argmax(v)) and ids[0] == np.argmax(v)
victim = ShaderProperty(mgr, 64)                         # 256 bytes: the allocation's padded end is the buffer's end
ptr, nbytes = C.c_void_p(), C.c_uint64()
call("nfai_hip_buf_info", mgr.handle, victim.buffer.handle, C.byref(ptr), C.byref(nbytes))
alias = _lib.H()
call("nfai_hip_buf_wrap", mgr.handle, ptr, nbytes.value + 64, C.byref(alias))
junk = np.full(nbytes.value + 64, 7, np.uint8)
call("nfai_hip_buf_upload", mgr.handle, alias, 0, junk.ctypes.data_as(C.c_void_p), junk.size)   # 64 bytes past the end
try:
    call("nfai_hip_ctx_synchronize", mgr.handle)
    print("MISSED")
except NfaiHipError as e:
    print("CAUGHT", e)
""" % ROOT
    env = {**os.environ, "NFAI_HIP_DEBUG_CANARY": "1"}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "CAUGHT" in r.stdout and "write outside buffer" in r.stdout and "after the buffer" in r.stdout, r.stdout

