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
