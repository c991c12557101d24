"""Synthetic Llama-3 shaped models (there is no network for checkpoints): SURVEY.md §8d recipe.

Weights ~ N(0, 0.02^2) rounded to fp16, norm gains 1 + N(0, 0.1^2) fp32, NumPy PCG64 seeded.
Tensor names and [N][K] row-major layout are the GGUF ones the reference looks up
(NFAI.Vulkan.Shaders/TransformerBlock.cs:41-101, NFAI.Models.Llama3/LlamaModel.cs:43,58).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict

import numpy as np


@dataclass(frozen=True)
class LlamaDims:
    name: str
    E: int
    L: int
    H: int
    Hkv: int
    D: int
    F: int
    V: int
    tied: bool = True

    def shapes(self) -> dict:
        """GGUF tensor name -> (N, K) for matrices / (E,) for norm gains."""
        s = {"token_embd.weight": (self.V, self.E), "output_norm.weight": (self.E,)}
        if not self.tied:
            s["output.weight"] = (self.V, self.E)
        for l in range(self.L):
            b = f"blk.{l}."
            s[b + "attn_norm.weight"] = (self.E,)
            s[b + "attn_q.weight"] = (self.H * self.D, self.E)
            s[b + "attn_k.weight"] = (self.Hkv * self.D, self.E)
            s[b + "attn_v.weight"] = (self.Hkv * self.D, self.E)
            s[b + "attn_output.weight"] = (self.E, self.H * self.D)
            s[b + "ffn_norm.weight"] = (self.E,)
            s[b + "ffn_gate.weight"] = (self.F, self.E)
            s[b + "ffn_up.weight"] = (self.F, self.E)
            s[b + "ffn_down.weight"] = (self.E, self.F)
        return s

    def n_params_read_per_token(self) -> int:
        """Matrix weights read once per decoded token (embedding table counted via lm_head when
        tied; the embedding row itself is E more)."""
        per_layer = (self.H * self.D * self.E + 2 * self.Hkv * self.D * self.E
                     + self.E * self.H * self.D + 3 * self.F * self.E)
        return self.L * per_layer + self.V * self.E

    def as_dict(self):
        return asdict(self)


# Public Llama-3 configurations.
LLAMA_32_1B = LlamaDims("llama-3.2-1b", 2048, 16, 32, 8, 64, 8192, 128256, True)
LLAMA_32_3B = LlamaDims("llama-3.2-3b", 3072, 28, 24, 8, 128, 8192, 128256, True)
LLAMA_31_8B = LlamaDims("llama-3.1-8b", 4096, 32, 32, 8, 128, 14336, 128256, False)
# Small shapes for CPU-sized parity runs (all K multiples of 256 so K-quants apply).
TINY = LlamaDims("tiny-llama", 256, 2, 4, 2, 64, 512, 512, True)
TINY_D128 = LlamaDims("tiny-llama-d128", 512, 3, 4, 2, 128, 1024, 768, False)

BY_NAME = {d.name: d for d in (LLAMA_32_1B, LLAMA_32_3B, LLAMA_31_8B, TINY, TINY_D128)}


def make_weights(dims: LlamaDims, seed: int = 1234, std: float = 0.02) -> dict:
    """name -> ndarray; matrices float16, gains float32."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shape in dims.shapes().items():
        if len(shape) == 1:
            out[name] = (1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        else:
            out[name] = (std * rng.standard_normal(shape, dtype=np.float32)).astype(np.float16)
    return out


def make_tokens(dims: LlamaDims, n: int, seed: int = 99) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, dims.V, size=n, dtype=np.uint32)


def make_metadata(dims: LlamaDims, eps: float = 1e-5) -> dict:
    """The GGUF metadata keys LlamaModel reads (LlamaModel.cs:23-39)."""
    return {
        "general.architecture": "llama", "general.name": dims.name, "llama.block_count": dims.L,
        "llama.attention.head_count": dims.H, "llama.attention.head_count_kv": dims.Hkv,
        "llama.attention.key_length": dims.D, "llama.attention.value_length": dims.D,
        "llama.rope.dimension_count": dims.D, "llama.rope.freq_base": 500000.0,
        "llama.attention.layer_norm_rms_epsilon": eps, "llama.embedding_length": dims.E,
        "llama.feed_forward_length": dims.F, "llama.vocab_size": dims.V,
        "tokenizer.ggml.bos_token_id": 1, "tokenizer.ggml.eos_token_id": 2,
    }
