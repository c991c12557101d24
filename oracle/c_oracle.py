"""ctypes binding of oracle/nfai_oracle.c (the CPU restatement).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnfai_oracle.so")
_SRC = os.path.join(_HERE, "nfai_oracle.c")

__all__ = [
    "build", "lib", "embed", "rmsnorm", "gemv", "gemv_f16w", "rope_freqs", "rope", "attn_scores",
    "attn_softmax", "attn_wsum", "silu", "mul", "add", "argmax", "topp", "widen_f16", "narrow_f16",
    "dequant_q4k", "dequant_q6k", "quantize_q4k", "quantize_q6k", "LlamaDesc", "OracleLlama",
    "num_threads", "set_num_threads", "Q4K_BYTES", "Q6K_BYTES", "QK_K",
]

QK_K, Q4K_BYTES, Q6K_BYTES = 256, 144, 210


def build(force: bool = False) -> str:
    """Compile nfai_oracle.c -> libnfai_oracle.so (gcc, see Makefile).  Building the checker is
    not using it."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnfai_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def effective_cpus() -> int:
    """CPUs this process may really use: min(affinity mask, cgroup quota).  On a shared GPU host
    os.cpu_count() is the whole machine while the container gets a slice; an OpenMP team sized to
    the machine spins on a few cores and makes every parallel region milliseconds long."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, min(n, 64))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        os.environ.setdefault("OMP_NUM_THREADS", str(effective_cpus()))
        L = C.CDLL(_SO)
        f32p, u16p, u8p, u32, f32 = (C.POINTER(C.c_float), C.POINTER(C.c_uint16),
                                      C.POINTER(C.c_uint8), C.c_uint32, C.c_float)
        sz = C.c_size_t
        sigs = {
            "orc_half_to_float": (f32, [C.c_uint16]),
            "orc_float_to_half": (C.c_uint16, [f32]),
            "orc_widen_f16": (None, [u16p, f32p, sz]),
            "orc_narrow_f16": (None, [f32p, u16p, sz]),
            "orc_embed": (None, [f32p, u32, u32, f32p]),
            "orc_rmsnorm": (None, [f32p, f32p, f32p, u32, f32]),
            "orc_gemv": (None, [f32p, f32p, f32p, u32, u32]),
            "orc_gemv_f16w": (None, [u16p, f32p, f32p, u32, u32]),
            "orc_rope_freqs": (None, [f32p, u32, f32, u32]),
            "orc_rope": (None, [f32p, f32p, f32p, u32, u32, u32, u32]),
            "orc_attn_scores": (None, [f32p, f32p, f32p, u32, u32, u32, u32]),
            "orc_attn_softmax": (None, [f32p, f32p, u32, u32, f32]),
            "orc_attn_wsum": (None, [f32p, f32p, f32p, u32, u32, u32, u32]),
            "orc_silu": (None, [f32p, f32p, u32]),
            "orc_mul": (None, [f32p, f32p, f32p, u32]),
            "orc_add": (None, [f32p, f32p, f32p, u32]),
            "orc_argmax": (u32, [f32p, u32]),
            "orc_topp": (u32, [f32p, u32, f32, f32, u32, f32, C.POINTER(C.c_uint32), f32p, C.POINTER(C.c_uint32)]),
            "orc_dequant_q4k": (None, [u8p, f32p, sz]),
            "orc_dequant_q6k": (None, [u8p, f32p, sz]),
            "orc_quantize_q4k": (None, [f32p, u8p, sz]),
            "orc_quantize_q6k": (None, [f32p, u8p, sz]),
            "orc_llama_create": (C.c_void_p, [C.c_void_p]),
            "orc_llama_destroy": (None, [C.c_void_p]),
            "orc_llama_set_globals": (None, [C.c_void_p, C.c_void_p, C.c_void_p, f32p]),
            "orc_llama_set_layer": (None, [C.c_void_p, u32, f32p] + [C.c_void_p] * 4 + [f32p]
                                    + [C.c_void_p] * 3),
            "orc_llama_reset": (None, [C.c_void_p]),
            "orc_llama_pos": (u32, [C.c_void_p]),
            "orc_llama_kcache": (f32p, [C.c_void_p, u32]),
            "orc_llama_vcache": (f32p, [C.c_void_p, u32]),
            "orc_llama_layers": (None, [C.c_void_p, f32p, u32, u32]),
            "orc_llama_step": (C.c_int, [C.c_void_p, u32, f32p]),
            "orc_llama_hidden": (f32p, [C.c_void_p]),
            "orc_llama_normed": (f32p, [C.c_void_p]),
            "orc_llama_advance": (None, [C.c_void_p]),
            "orc_num_threads": (C.c_int, []),
            "orc_set_num_threads": (None, [C.c_int]),
        }
        for name, (res, args) in sigs.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        L.orc_set_num_threads(min(effective_cpus(), max(1, L.orc_num_threads())))
        _lib = L
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray, ty=C.c_float):
    return a.ctypes.data_as(C.POINTER(ty))


def num_threads() -> int:
    return lib().orc_num_threads()


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(n)


def widen_f16(h: np.ndarray) -> np.ndarray:
    h = np.ascontiguousarray(h).view(np.uint16)
    out = np.empty(h.shape, np.float32)
    lib().orc_widen_f16(_p(h, C.c_uint16), _p(out), h.size)
    return out


def narrow_f16(f: np.ndarray) -> np.ndarray:
    f = _f32(f)
    out = np.empty(f.shape, np.uint16)
    lib().orc_narrow_f16(_p(f), _p(out, C.c_uint16), f.size)
    return out.view(np.float16)


def embed(emb, tok: int) -> np.ndarray:
    emb = _f32(emb)
    E = emb.shape[1]
    out = np.empty(E, np.float32)
    lib().orc_embed(_p(emb), tok, E, _p(out))
    return out


def rmsnorm(x, g, eps: float) -> np.ndarray:
    x, g = _f32(x), _f32(g)
    y = np.empty_like(x)
    lib().orc_rmsnorm(_p(x), _p(g), _p(y), x.size, eps)
    return y


def gemv(W, x) -> np.ndarray:
    W, x = _f32(W), _f32(x)
    N, K = W.shape
    y = np.empty(N, np.float32)
    lib().orc_gemv(_p(W), _p(x), _p(y), N, K)
    return y


def gemv_f16w(W16, x) -> np.ndarray:
    W16 = np.ascontiguousarray(W16).view(np.uint16)
    x = _f32(x)
    N, K = W16.shape
    y = np.empty(N, np.float32)
    lib().orc_gemv_f16w(_p(W16, C.c_uint16), _p(x), _p(y), N, K)
    return y


def rope_freqs(rope_dims: int, base: float = 500000.0, n_valid: int | None = None) -> np.ndarray:
    f = np.zeros(rope_dims // 2, np.float32)
    lib().orc_rope_freqs(_p(f), rope_dims, base, rope_dims // 2 if n_valid is None else n_valid)
    return f


def rope(x, freqs, rope_dims: int, n_heads: int, head_dim: int, pos: int) -> np.ndarray:
    x, freqs = _f32(x), _f32(freqs)
    y = np.empty_like(x)
    lib().orc_rope(_p(x), _p(y), _p(freqs), rope_dims, n_heads, head_dim, pos)
    return y


def attn_scores(q, Kc, H: int, Hkv: int, D: int, S: int) -> np.ndarray:
    q, Kc = _f32(q), _f32(Kc)
    s = np.empty(H * S, np.float32)
    lib().orc_attn_scores(_p(q), _p(Kc), _p(s), H, Hkv, D, S)
    return s.reshape(H, S)


def attn_softmax(s, eps: float = 1e-5) -> np.ndarray:
    s = _f32(s)
    H, S = s.shape
    w = np.empty_like(s)
    lib().orc_attn_softmax(_p(s), _p(w), H, S, eps)
    return w


def attn_wsum(w, Vc, H: int, Hkv: int, D: int, S: int) -> np.ndarray:
    w, Vc = _f32(w), _f32(Vc)
    o = np.empty(H * D, np.float32)
    lib().orc_attn_wsum(_p(w), _p(Vc), _p(o), H, Hkv, D, S)
    return o


def silu(x) -> np.ndarray:
    x = _f32(x)
    y = np.empty_like(x)
    lib().orc_silu(_p(x), _p(y), x.size)
    return y


def mul(a, b) -> np.ndarray:
    a, b = _f32(a), _f32(b)
    y = np.empty_like(a)
    lib().orc_mul(_p(a), _p(b), _p(y), a.size)
    return y


def add(a, b) -> np.ndarray:
    a, b = _f32(a), _f32(b)
    y = np.empty_like(a)
    lib().orc_add(_p(a), _p(b), _p(y), a.size)
    return y


def argmax(v) -> int:
    v = _f32(v)
    return int(lib().orc_argmax(_p(v), v.size))


def topp(values, temperature: float = 0.5, topP: float = 0.95, topK: int = 40, rand: float = 0.0):
    """SamplingUtils.TopP (SamplingUtils.cs:5-33) with the random draw given: (token, ids[topK], probs[topK], n_kept)."""
    v = _f32(values)
    k = min(int(topK), v.size)
    ids = np.empty(k, np.uint32)
    probs = np.empty(k, np.float32)
    kept = C.c_uint32()
    tok = lib().orc_topp(_p(v), v.size, float(temperature), float(topP), k, float(rand),
                         ids.ctypes.data_as(C.POINTER(C.c_uint32)), _p(probs), C.byref(kept))
    return int(tok), ids, probs, int(kept.value)


def dequant_q4k(blocks: np.ndarray, n_weights: int) -> np.ndarray:
    blocks = np.ascontiguousarray(blocks, np.uint8)
    out = np.empty(n_weights, np.float32)
    lib().orc_dequant_q4k(_p(blocks, C.c_uint8), _p(out), n_weights)
    return out


def dequant_q6k(blocks: np.ndarray, n_weights: int) -> np.ndarray:
    blocks = np.ascontiguousarray(blocks, np.uint8)
    out = np.empty(n_weights, np.float32)
    lib().orc_dequant_q6k(_p(blocks, C.c_uint8), _p(out), n_weights)
    return out


def quantize_q4k(w) -> np.ndarray:
    w = _f32(w).reshape(-1)
    assert w.size % QK_K == 0
    out = np.empty(w.size // QK_K * Q4K_BYTES, np.uint8)
    lib().orc_quantize_q4k(_p(w), _p(out, C.c_uint8), w.size)
    return out


def quantize_q6k(w) -> np.ndarray:
    w = _f32(w).reshape(-1)
    assert w.size % QK_K == 0
    out = np.empty(w.size // QK_K * Q6K_BYTES, np.uint8)
    lib().orc_quantize_q6k(_p(w), _p(out, C.c_uint8), w.size)
    return out


class _Desc(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("E", "L", "H", "Hkv", "D", "F", "V", "C")] + [
        ("eps", C.c_float), ("rope_base", C.c_float), ("rope_dims", C.c_uint32),
        ("rope_n_freqs", C.c_uint32), ("weights_f16", C.c_uint32)]


@dataclass
class LlamaDesc:
    E: int
    L: int
    H: int
    Hkv: int
    D: int
    F: int
    V: int
    C: int
    eps: float = 1e-5
    rope_base: float = 500000.0
    rope_dims: int | None = None
    rope_n_freqs: int | None = None  # None = spec-correct full table; 32 = reference truncation

    def resolved(self):
        rd = self.D if self.rope_dims is None else self.rope_dims
        nf = rd // 2 if self.rope_n_freqs is None else self.rope_n_freqs
        return rd, nf


class OracleLlama:
    """Whole-model CPU restatement (LlamaModel.cs:21-68,99-174 + TransformerBlock.cs:127-184).

    ``weights``: dict name -> ndarray using GGUF tensor names (token_embd.weight,
    blk.N.attn_q.weight ... output_norm.weight, optional output.weight).  Matrices may be
    float16 (kept as fp16, widened on the fly — identical operand values) or float32.
    """

    MATS = ("attn_q", "attn_k", "attn_v", "attn_output", "ffn_gate", "ffn_up", "ffn_down")

    def __init__(self, desc: LlamaDesc, weights: dict):
        self.desc = desc
        rd, nf = desc.resolved()
        emb = weights["token_embd.weight"]
        f16 = emb.dtype == np.float16
        self._keep = {}

        def mat(name):
            a = weights[name]
            a = np.ascontiguousarray(a, np.float16 if f16 else np.float32)
            self._keep[name] = a
            return a.ctypes.data_as(C.c_void_p)

        def vec(name):
            a = _f32(weights[name])
            self._keep[name] = a
            return _p(a)

        d = _Desc(desc.E, desc.L, desc.H, desc.Hkv, desc.D, desc.F, desc.V, desc.C, desc.eps,
                  desc.rope_base, rd, nf, 1 if f16 else 0)
        L = lib()
        self._h = C.c_void_p(L.orc_llama_create(C.byref(d)))
        out = mat("output.weight") if "output.weight" in weights else None
        L.orc_llama_set_globals(self._h, mat("token_embd.weight"), out, vec("output_norm.weight"))
        for l in range(desc.L):
            b = f"blk.{l}."
            L.orc_llama_set_layer(self._h, l, vec(b + "attn_norm.weight"), mat(b + "attn_q.weight"),
                                  mat(b + "attn_k.weight"), mat(b + "attn_v.weight"),
                                  mat(b + "attn_output.weight"), vec(b + "ffn_norm.weight"),
                                  mat(b + "ffn_gate.weight"), mat(b + "ffn_up.weight"),
                                  mat(b + "ffn_down.weight"))

    def close(self):
        if self._h:
            lib().orc_llama_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def pos(self) -> int:
        return lib().orc_llama_pos(self._h)

    def reset(self):
        lib().orc_llama_reset(self._h)

    def step(self, tok: int, want_logits: bool = True):
        logits = np.empty(self.desc.V, np.float32) if want_logits else None
        rc = lib().orc_llama_step(self._h, int(tok), _p(logits) if want_logits else None)
        if rc != 0:
            raise RuntimeError("KV cache capacity exceeded")
        return logits

    def hidden(self) -> np.ndarray:
        return np.ctypeslib.as_array(lib().orc_llama_hidden(self._h), (self.desc.E,)).copy()

    def normed(self) -> np.ndarray:
        return np.ctypeslib.as_array(lib().orc_llama_normed(self._h), (self.desc.E,)).copy()

    def layers(self, hidden: np.ndarray, l0: int, l1: int) -> np.ndarray:
        h = _f32(hidden).copy()
        lib().orc_llama_layers(self._h, _p(h), l0, l1)
        return h

    def advance(self):
        lib().orc_llama_advance(self._h)

    def kcache(self, l: int) -> np.ndarray:
        d = self.desc
        return np.ctypeslib.as_array(lib().orc_llama_kcache(self._h, l), (d.C, d.Hkv * d.D))

    def vcache(self, l: int) -> np.ndarray:
        d = self.desc
        return np.ctypeslib.as_array(lib().orc_llama_vcache(self._h, l), (d.C, d.Hkv * d.D))

    def greedy(self, prompt, n_new: int):
        """Prompt one token at a time (LlamaModel.cs:103-126), then greedy ArgMax feedback
        (SamplingUtils.cs:43-57 in place of the stochastic TopP).  Returns (tokens, logits list)."""
        logits = None
        for t in prompt:
            logits = self.step(t)
        toks, all_logits = [], []
        for _ in range(n_new):
            all_logits.append(logits)
            t = argmax(logits)
            toks.append(t)
            logits = self.step(t)
        return toks, all_logits
