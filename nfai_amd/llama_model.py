"""LlamaModel / LlamaModelFactory / ModelOptions / SamplingUtils on the HIP backend.

Mirrors NFAI.Models.Llama3/LlamaModel.cs, LlamaModelFactory.cs, SamplingUtils.cs and
NFAI.Models/ModelOptions.cs.  `LlamaModel` drives the C++ model object of libnfai_hip.so
(fused kernels, one hipGraph per token); `ChainLlamaModel` builds the same network out of the
1:1 operator classes of `nfai_amd.shaders` exactly as the reference constructor wires them
(LlamaModel.cs:43-67) and is the op-surface parity harness.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import call
from .hip import HipBufferManager
from .shaders import MatrixMultiplyShader, RMSNormShader, TokenEmbedShader, TransformerBlock

_GGML = {np.dtype(np.float32): _lib.F32, np.dtype(np.float16): _lib.F16}


@dataclass
class ModelOptions:
    """≙ NFAI.Models/ModelOptions.cs:3-8."""
    GGUFPath: str = ""
    KVCacheSize: int = 512


@dataclass
class QuantTensor:
    """A GGUF tensor kept in its on-disk block encoding (Q4_K / Q6_K): raw bytes + logical shape."""
    data: np.ndarray  # uint8
    ggml_type: int
    shape: tuple  # (rows, cols) = (ne1, ne0)


def dims_from_metadata(metadata: dict, tensors: dict) -> dict:
    """The keys LlamaModel reads (LlamaModel.cs:23-39); F and V come from tensor shapes as the
    reference takes them from ComputeCollection.Shape (TransformerBlock.cs:47-101)."""
    eps = next((v for k, v in metadata.items() if "epsilon" in k), 0.0)  # first key containing "epsilon" (:28)
    H = int(metadata["llama.attention.head_count"])
    E = int(_shape(tensors["token_embd.weight"])[1])
    D = int(metadata.get("llama.attention.key_length", E // H))
    return dict(
        E=E, L=int(metadata["llama.block_count"]), H=H, Hkv=int(metadata["llama.attention.head_count_kv"]), D=D,
        F=int(_shape(tensors["blk.0.ffn_gate.weight"])[0]), V=int(_shape(tensors["token_embd.weight"])[0]),
        eps=float(eps), rope_dims=int(metadata.get("llama.rope.dimension_count", D)),
        rope_base=float(metadata.get("llama.rope.freq_base", 500000.0)))


def _shape(t):
    return t.shape if isinstance(t, (np.ndarray, QuantTensor)) else t.shape


class SamplingUtils:
    """≙ SamplingUtils.cs:3-58 (host side, NumPy)."""

    @staticmethod
    def Softmax(logits: np.ndarray) -> np.ndarray:
        e = np.exp(logits - logits.max(), dtype=np.float32)
        return e / e.sum(dtype=np.float32)

    @staticmethod
    def TopP(values: np.ndarray, temperature: float = 0.5, topP: float = 0.95, topK: int = 40,
             rng: np.random.Generator | None = None) -> int:
        probs = SamplingUtils.Softmax(np.asarray(values, np.float32) / np.float32(temperature))
        order = np.argsort(-probs, kind="stable")[:topK]
        cum = np.cumsum(probs[order], dtype=np.float32)
        keep = int(np.searchsorted(cum, topP, side="left")) + 1  # include the element that crosses topP
        order = order[:keep]
        p = probs[order] / probs[order].sum(dtype=np.float32)
        r = (rng or np.random.default_rng()).random(dtype=np.float32)
        running = np.cumsum(p, dtype=np.float32)
        idx = int(np.searchsorted(running, r, side="right"))
        return int(order[min(idx, len(order) - 1)])

    @staticmethod
    def TopPFromCandidates(ids: np.ndarray, probs: np.ndarray, topP: float = 0.95, rand: float | None = None,
                           rng: np.random.Generator | None = None) -> int:
        """SamplingUtils.cs:14-31 on the (index, probability) pairs `.Take(topK)` leaves (:13) — what
        nfai_hip_llama_decode_topk returns: nucleus cut (the element that crosses topP is kept), renormalise, draw.
        `rand` stands in for Random.Shared.NextSingle() (:24).  Vectorised: np.cumsum over float32 adds in sequence, exactly the
        reference's running sums (TopPFromCandidatesLoop is the statement-by-statement form; tests compare the two)."""
        p = np.asarray(probs, np.float32)
        cum = np.cumsum(p, dtype=np.float32)                        # cumulative += prob (:17-18)
        crossed = np.flatnonzero(cum >= np.float32(topP))
        keep = int(crossed[0]) + 1 if crossed.size else p.size      # the element that crosses topP is kept (:19-20)
        total = np.float32(np.sum(p[:keep], dtype=np.float64))      # Enumerable.Sum over floats accumulates in double (:23)
        r = np.float32(rand) if rand is not None else (rng or np.random.default_rng()).random(dtype=np.float32)
        running = np.cumsum(p[:keep] / total, dtype=np.float32)     # running += prob / total (:27-28)
        hit = np.flatnonzero(r < running)
        return int(ids[int(hit[0]) if hit.size else keep - 1])

    @staticmethod
    def TopPFromCandidatesLoop(ids: np.ndarray, probs: np.ndarray, topP: float = 0.95, rand: float = 0.0) -> int:
        """The same, one reference statement per line (SamplingUtils.cs:14-31)."""
        cumulative = np.float32(0.0)
        keep = 0
        for p in probs:
            cumulative = np.float32(cumulative + np.float32(p))
            keep += 1
            if cumulative >= np.float32(topP):
                break
        total = np.float32(np.sum(probs[:keep], dtype=np.float64))
        r = np.float32(rand)
        running = np.float32(0.0)
        for i in range(keep):
            running = np.float32(running + np.float32(probs[i]) / total)
            if r < running:
                return int(ids[i])
        return int(ids[keep - 1])

    @staticmethod
    def ArgMax(values: np.ndarray) -> int:
        return int(np.argmax(values))  # first maximum, as values.ToList().IndexOf(max) (:56)


class LlamaModel:
    """≙ LlamaModel (LlamaModel.cs:10-175) for blocks [layer_begin, layer_end) of the network.

    tensors: dict GGUF-name -> ndarray (float16/float32) | QuantTensor | (device_ptr:int, ggml_type, rows, cols).
    """

    def __init__(self, mgr: HipBufferManager, metadata: dict, tensors: dict, contextSize: int = 1024, *,
                 tokenizer=None, unfused: bool = False, graph: bool = True, kv_f16: bool = False,
                 rope_n_freqs: int | None = None, rope_base: float | None = 500000.0,
                 layer_range: tuple[int, int] | None = None, dims: dict | None = None, max_batch: int = 0,
                 share_from: "LlamaModel | None" = None, engine: bool | None = None, prompt_prefill: bool = True):
        self.mgr = mgr
        # RunAsync's prompt phase: True = all prompt tokens but the last through nfai_hip_llama_ingest (the batched MFMA prefill when
        # the model was created with max_batch > 0); False = token by token through the M = 1 path, bit for bit what the decode
        # path computes (parity runs against the reference's loop, LlamaModel.cs:103-126)
        self.promptPrefill = bool(prompt_prefill)
        d = dims or dims_from_metadata(metadata, tensors)
        self.dims = d
        self.ModelName = str(metadata.get("general.name", "unknown"))
        if tokenizer is None and "tokenizer.ggml.tokens" in metadata and "tokenizer.ggml.merges" in metadata:
            from .tokenizer import Tokenizer
            tokenizer = Tokenizer(metadata)  # LlamaModel.cs:41
        self.tokenizer = tokenizer
        self.firstInput = True
        lb, le = layer_range or (0, d["L"])
        flags = (_lib.LLAMA_UNFUSED if unfused else 0) | (0 if graph else _lib.LLAMA_NO_GRAPH) | (_lib.LLAMA_KV_F16 if kv_f16 else 0)
        if engine:  # the one-launch-per-block weight-streaming engine (fp16 models); None = the library's default
            flags |= _lib.LLAMA_ENGINE
        rd = d["rope_dims"]
        desc = _lib.LlamaDescC(d["E"], d["L"], d["H"], d["Hkv"], d["D"], d["F"], d["V"], int(contextSize), d["eps"],
                               # the reference ignores llama.rope.freq_base and uses 500000 (TransformerBlock.cs:33)
                               float(rope_base if rope_base is not None else d["rope_base"]), rd,
                               rd // 2 if rope_n_freqs is None else rope_n_freqs, lb, le, flags, int(max_batch))
        self.C = int(contextSize)
        h = _lib.H()
        call("nfai_hip_llama_create", mgr.handle, C.byref(desc), C.byref(h))
        self.handle = h
        self._keep = []
        if share_from is not None:  # another slot of the same pipeline stage: the donor's weights, no copy, no second repack
            self._donor = share_from
            call("nfai_hip_llama_share_tensors", self.handle, share_from.handle)
        else:
            for name, t in tensors.items():
                self.SetTensor(name, t)
        call("nfai_hip_llama_finalize", self.handle)

    def SetTensor(self, name: str, t) -> None:
        if isinstance(t, tuple):  # already resident in HBM
            ptr, ty, rows, cols = t
            call("nfai_hip_llama_set_tensor_device", self.handle, name.encode(), ty, rows, cols, C.c_void_p(ptr))
            return
        if isinstance(t, QuantTensor):
            a, ty, (rows, cols) = np.ascontiguousarray(t.data), t.ggml_type, t.shape
        else:
            a = np.ascontiguousarray(t)
            ty = _GGML[a.dtype]
            rows, cols = (1, a.shape[0]) if a.ndim == 1 else a.shape
        call("nfai_hip_llama_set_tensor", self.handle, name.encode(), ty, rows, cols, a.ctypes.data_as(C.c_void_p))

    # -- one token (LlamaModel.cs:116-125)
    def Step(self, token: int, want_logits: bool = True):
        logits = np.empty(self.dims["V"], np.float32) if want_logits else None
        am = C.c_uint32()
        call("nfai_hip_llama_decode_step", self.handle, int(token),
             logits.ctypes.data_as(C.POINTER(C.c_float)) if want_logits else None, C.byref(am))
        return logits, am.value

    def StepTopK(self, token: int, temperature: float = 0.5, topK: int = 40):
        """One token, then the candidates of SamplingUtils.TopP formed on the device (SamplingUtils.cs:5-13): (ids[topK],
        probs[topK]); 8*topK + 8 bytes come back instead of V floats."""
        ids = np.empty(topK, np.uint32)
        probs = np.empty(topK, np.float32)
        call("nfai_hip_llama_decode_topk", self.handle, int(token), float(temperature), int(topK),
             ids.ctypes.data_as(C.POINTER(C.c_uint32)), probs.ctypes.data_as(C.POINTER(C.c_float)))
        return ids, probs

    def Greedy(self, first_token: int, n_steps: int) -> np.ndarray:
        out = np.empty(n_steps, np.uint32)
        call("nfai_hip_llama_decode_greedy", self.handle, int(first_token), n_steps, out.ctypes.data_as(C.POINTER(C.c_uint32)))
        return out

    def SetToken(self, token: int) -> None:
        call("nfai_hip_llama_set_token", self.handle, int(token))

    def Enqueue(self, n_steps: int) -> None:
        call("nfai_hip_llama_decode_enqueue", self.handle, n_steps)

    def FetchTokens(self, n: int) -> np.ndarray:
        out = np.empty(n, np.uint32)
        call("nfai_hip_llama_fetch_tokens", self.handle, n, out.ctypes.data_as(C.POINTER(C.c_uint32)))
        return out

    def Prefill(self, tokens, want_logits: bool = True):
        t = np.ascontiguousarray(tokens, np.uint32)
        logits = np.empty(self.dims["V"], np.float32) if want_logits else None
        call("nfai_hip_llama_prefill", self.handle, t.ctypes.data_as(C.POINTER(C.c_uint32)), t.size,
             logits.ctypes.data_as(C.POINTER(C.c_float)) if want_logits else None)
        return logits

    def Ingest(self, tokens) -> None:
        """Prompt tokens whose output is never sampled (LlamaModel.cs:103-126 keeps only the last token's logits): K / V rows only."""
        t = np.ascontiguousarray(tokens, np.uint32)
        if t.size:
            call("nfai_hip_llama_ingest", self.handle, t.ctypes.data_as(C.POINTER(C.c_uint32)), t.size)

    def StageStep(self, token: int = 0, hidden_in: int | None = None, hidden_out: int | None = None, want_logits: bool = False,
                  want_argmax: bool = False):
        logits = np.empty(self.dims["V"], np.float32) if want_logits else None
        am = C.c_uint32()
        call("nfai_hip_llama_stage_step", self.handle, int(token), C.c_void_p(hidden_in), C.c_void_p(hidden_out),
             logits.ctypes.data_as(C.POINTER(C.c_float)) if want_logits else None, C.byref(am) if (want_argmax or want_logits) else None)
        return logits, am.value

    def TokenToDevice(self, dst_ptr: int) -> None:
        call("nfai_hip_llama_token_to_device", self.handle, C.c_void_p(dst_ptr))

    def TokenFromDevice(self, src_ptr: int) -> None:
        call("nfai_hip_llama_token_from_device", self.handle, C.c_void_p(src_ptr))

    def Reset(self) -> None:
        call("nfai_hip_llama_reset", self.handle)

    def SetPos(self, pos: int) -> None:
        call("nfai_hip_llama_set_pos", self.handle, int(pos))

    @property
    def Pos(self) -> int:
        p = C.c_uint32()
        call("nfai_hip_llama_pos", self.handle, C.byref(p))
        return p.value

    def Read(self, which: int, n: int) -> np.ndarray:
        out = np.empty(n, np.float32)
        call("nfai_hip_llama_read", self.handle, which, out.ctypes.data_as(C.POINTER(C.c_float)), n)
        return out

    def ReadKV(self, layer: int, is_v: bool, pos: int) -> np.ndarray:
        out = np.empty(self.dims["Hkv"] * self.dims["D"], np.float32)
        call("nfai_hip_llama_read_kv", self.handle, layer, int(is_v), pos, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def BytesPerToken(self, pos: int) -> tuple[int, int]:
        t, d = C.c_uint64(), C.c_uint64()
        call("nfai_hip_llama_bytes_per_token", self.handle, pos, C.byref(t), C.byref(d))
        return t.value, d.value

    def ProfileStep(self, token: int):
        ms = (C.c_float * 8)()
        n = (C.c_uint32 * 8)()
        call("nfai_hip_llama_profile_step", self.handle, int(token), ms, n)
        names = ["qkv", "attn", "wo", "gateup", "down", "lmhead", "other", "engine"]
        return {k: (ms[i], n[i]) for i, k in enumerate(names)}

    def ProfileKernel(self, token: int, kernel_class: str, reps: int = 4) -> float:
        """Average duration in microseconds of one kernel class: its launches of a step replayed back to back, `reps` rounds."""
        names = ["qkv", "attn", "wo", "gateup", "down", "lmhead", "other", "engine"]
        us = C.c_float()
        call("nfai_hip_llama_profile_kernel", self.handle, int(token), names.index(kernel_class), int(reps), C.byref(us))
        return us.value

    # -- the token loop (LlamaModel.RunAsync, :99-174)
    def RunAsync(self, prompt: str, greedy: bool = False, max_tokens: int | None = None, rng=None):
        if self.tokenizer is None:
            raise RuntimeError("RunAsync needs a tokenizer (nfai_amd.tokenizer.Tokenizer(metadata))")
        tokenIds = self.tokenizer.Tokenize(prompt, addBos=self.firstInput)
        self.firstInput = False
        # greedy: ArgMax on the device (SamplingUtils.cs:43-57).  Otherwise the reference's default (LlamaModel.cs:130,165):
        # TopP(temperature 0.5, topP 0.95, topK 40) — the softmax over V and the top-40 are formed on the device
        # (nfai_hip_llama_decode_topk), the nucleus cut and the draw here; the V logits never cross PCIe.
        def step(tok):
            if greedy:
                return self.Step(tok, want_logits=False)[1]
            ids, probs = self.StepTopK(tok)
            return SamplingUtils.TopPFromCandidates(ids, probs, rng=rng)
        # prompt (:103-126): only the last token's output is sampled (:128-130), so the tokens in front of it only have to fill the
        # KV cache — ONE call, the MFMA prefill in chunks of max_batch (fp16 operands; INTEGRATION.md 3 states the precision trade);
        # promptPrefill = False feeds them one at a time as the reference does
        if self.promptPrefill:
            self.Ingest(tokenIds[:-1])
        else:
            for tok in tokenIds[:-1]:
                self.Step(tok, want_logits=False)
        tk = step(tokenIds[-1])
        yield self.tokenizer.Detokenize([tk])
        n = 1
        while tk != self.tokenizer.EosTokenId and (max_tokens is None or n < max_tokens):
            tk = step(tk)
            n += 1
            if tk != self.tokenizer.EosTokenId:
                yield self.tokenizer.Detokenize([tk])

    def Dispose(self) -> None:  # the reference throws NotImplementedException (:70-74)
        if self.handle is not None:
            call("nfai_hip_llama_destroy", self.handle)
            self.handle = None


class LlamaModelFactory:
    """≙ LlamaModelFactory (LlamaModelFactory.cs:7-45): the plugin hook AbstractModelFactory.TryCreate."""

    def __init__(self, device: int = 0):
        self.mgr = HipBufferManager(device)

    PROMPT_CHUNK = 512  # tokens per MFMA prefill chunk of the provider path (workspace ~ 60 KB per token at 3B)

    def TryCreate(self, metadata: dict, tensors: dict, modelOptions: ModelOptions, **kw):
        if str(metadata.get("general.architecture", "")) != "llama":
            return False, None
        # the provider path ingests prompts through the MFMA prefill: give the model its workspace unless the caller decides otherwise
        kw.setdefault("max_batch", min(int(modelOptions.KVCacheSize), self.PROMPT_CHUNK))
        return True, LlamaModel(self.mgr, metadata, tensors, modelOptions.KVCacheSize, **kw)

    def Dispose(self) -> None:
        self.mgr.Dispose()


class ChainLlamaModel:
    """The network as the reference constructor builds it out of op objects (LlamaModel.cs:43-67):
    TokenEmbedShader -> TransformerBlock x L -> RMSNormShader -> MatrixMultiplyShader(lm_head, tied)."""

    def __init__(self, mgr, metadata: dict, tensors: dict, contextSize: int = 1024, ropeTableEntries: int | None = 32):
        d = dims_from_metadata(metadata, tensors)
        self.dims, self.mgr = d, mgr
        emb = tensors["token_embd.weight"]
        self.embedShader = TokenEmbedShader(mgr, 1, emb.shape[1], emb)
        lastProp = self.embedShader.GetOutputProperty()
        self.transformerBlocks = []
        for i in range(d["L"]):
            blk = TransformerBlock(mgr, tensors, d["D"], d["H"], d["Hkv"], contextSize, d["eps"], i, d["rope_base"],
                                   d["rope_dims"], i, ropeTableEntries)
            self.transformerBlocks.append(blk)
            blk.GetInputProperty().BindShaderProprty(lastProp)
            lastProp = blk.GetOutputProperty()
        self.outputNormLayer = RMSNormShader(mgr, d["E"], tensors["output_norm.weight"], d["eps"])
        self.outputNormLayer.GetInputProperty().BindShaderProprty(lastProp)
        lastProp = self.outputNormLayer.GetOutputProperty()
        self.lmHead = MatrixMultiplyShader(mgr, 1, emb.shape[1], emb.shape[0], None)
        self.lmHead.GetInputProperty().BindShaderProprty(lastProp)
        self.lmHead.GetWeightProperty().BindShaderProprty(self.embedShader.GetWeightProperty())  # always tied (:64-67)

    def Step(self, token: int) -> np.ndarray:
        self.embedShader.Compute(token)
        for blk in self.transformerBlocks:
            blk.Compute()
        self.outputNormLayer.Compute()
        self.lmHead.Compute()
        return self.lmHead.GetOutputs()
