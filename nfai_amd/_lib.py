"""ctypes binding of libnfai_hip.so (include/nfai_hip.h).  Fails loudly: no CPU fallback exists."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("NFAI_HIP_LIB") or os.path.join(_HERE, "csrc", "libnfai_hip.so")  # NFAI_HIP_LIB: diagnostic builds (tools/)

OK, ERR_INVALID, ERR_HIP, ERR_OOM, ERR_KV_FULL, ERR_UNSUPPORTED, ERR_STATE = range(7)
F32, F16, Q4_K, Q6_K = 0, 1, 12, 14
LLAMA_UNFUSED, LLAMA_NO_GRAPH, LLAMA_KV_F16, LLAMA_PREFETCH, LLAMA_ENGINE = 1, 2, 4, 8, 16
TOKEN_ON_DEVICE = 0xFFFFFFFF


class NfaiHipError(RuntimeError):
    """Raised for every non-zero status (the reference throws on any non-Success VkResult,
    NFAI.Vulkan/VulkanBufferManager.cs:61-87)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"nfai_hip status {code}: {msg}")
        self.code = code


class KVCacheFull(NfaiHipError):
    pass


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 64), ("total_mem_bytes", C.c_uint64),
                ("compute_units", C.c_uint32), ("wavefront_size", C.c_uint32),
                ("lds_bytes_per_cu", C.c_uint32), ("clock_khz", C.c_uint32)]


class LlamaDescC(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("E", "L", "H", "Hkv", "D", "F", "V", "C")] + [
        ("eps", C.c_float), ("rope_base", C.c_float), ("rope_dims", C.c_uint32),
        ("rope_n_freqs", C.c_uint32), ("layer_begin", C.c_uint32), ("layer_end", C.c_uint32),
        ("flags", C.c_uint32), ("max_batch", C.c_uint32)]


class PpOp(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("count", C.c_uint32), ("peer", C.c_uint32), ("kind", C.c_uint32), ("reserved", C.c_uint32)]


u64, u32, i32, f32, vp = C.c_uint64, C.c_uint32, C.c_int32, C.c_float, C.c_void_p
H = u64  # handles

# name -> argtypes; every function returns int32 status except last_error / abi_version
SIGNATURES = {
    "nfai_hip_ctx_create": [i32, C.POINTER(H)],
    "nfai_hip_ctx_create_on_stream": [i32, vp, C.POINTER(H)],
    "nfai_hip_ctx_destroy": [H],
    "nfai_hip_ctx_synchronize": [H],
    "nfai_hip_ctx_device_info": [H, C.POINTER(DeviceInfo)],
    "nfai_hip_ctx_xcd_shares": [H, C.POINTER(C.c_uint16), C.POINTER(f32)],
    "nfai_hip_timer_begin": [H],
    "nfai_hip_timer_end": [H, C.POINTER(f32)],
    "nfai_hip_buf_alloc": [H, u64, C.POINTER(H)],
    "nfai_hip_buf_wrap": [H, vp, u64, C.POINTER(H)],
    "nfai_hip_buf_free": [H, H],
    "nfai_hip_buf_upload": [H, H, u64, vp, u64],
    "nfai_hip_buf_download": [H, H, u64, vp, u64],
    "nfai_hip_buf_copy": [H, H, u64, H, u64, u64],
    "nfai_hip_buf_zero": [H, H],
    "nfai_hip_buf_info": [H, H, C.POINTER(vp), C.POINTER(u64)],
    "nfai_hip_weight_upload": [H, i32, u64, u64, vp, C.POINTER(H)],
    "nfai_hip_weight_bytes": [i32, u64, u64, C.POINTER(u64)],
    "nfai_hip_embed": [H, H, i32, H, H, u32],
    "nfai_hip_rmsnorm": [H, H, H, H, u32, f32],
    "nfai_hip_gemv": [H, H, i32, H, H, u64, u32, u32],
    "nfai_hip_gemm_f16": [H, H, H, H, H, u32, u32, u32, i32],
    "nfai_hip_gemm_f16_ex": [H, H, H, H, H, H, u32, u32, u32, i32, i32, u32, u32, u32, u32],
    "nfai_hip_attn_prefill": [H, H, H, H, H, u32, u32, u32, u32, u32, u32],
    "nfai_hip_gemm_kq": [H, H, H, i32, H, H, u32, u32, u32],
    "nfai_hip_rope": [H, H, u64, H, u64, H, u32, u32, u32, u32],
    "nfai_hip_attn_scores": [H, H, H, H, u32, u32, u32, u32],
    "nfai_hip_attn_softmax": [H, H, H, u32, u32, f32],
    "nfai_hip_attn_wsum": [H, H, H, H, u32, u32, u32, u32],
    "nfai_hip_silu": [H, H, H, u32],
    "nfai_hip_mul": [H, H, H, H, u32],
    "nfai_hip_add": [H, H, H, H, u32],
    "nfai_hip_argmax": [H, H, u32, H],
    "nfai_hip_topk": [H, H, u32, f32, u32, C.POINTER(u32), C.POINTER(f32)],
    "nfai_hip_attn_decode": [H, H, H, H, H, u32, u32, u32, u32, u32, i32],
    "nfai_hip_gemv_fused": [H, H, i32, H, H, f32, H, H, u32, u32],
    "nfai_hip_lmhead_argmax": [H, H, i32, H, H, f32, H, H, u32, u32],
    "nfai_hip_gemv_gateup_silu": [H, H, H, i32, H, H, f32, H, u32, u32],
    "nfai_hip_gemv_qkv_rope": [H, H, H, H, i32, H, H, f32, H, u32, H, H, H, u32, u32, u32, u32, i32, u32],
    "nfai_hip_engine_block": [H, H, H, H, H, H, H, H, f32, u32, u32, u32, H, H, H, H, H, u32, H, H, H, u32, u32, u32, u32, i32, H, H],
    "nfai_hip_llama_create": [H, C.POINTER(LlamaDescC), C.POINTER(H)],
    "nfai_hip_llama_destroy": [H],
    "nfai_hip_llama_set_tensor": [H, C.c_char_p, i32, u64, u64, vp],
    "nfai_hip_llama_set_tensor_device": [H, C.c_char_p, i32, u64, u64, vp],
    "nfai_hip_llama_finalize": [H],
    "nfai_hip_llama_share_tensors": [H, H],
    "nfai_hip_llama_decode_step": [H, u32, C.POINTER(f32), C.POINTER(u32)],
    "nfai_hip_llama_decode_topk": [H, u32, f32, u32, C.POINTER(u32), C.POINTER(f32)],
    "nfai_hip_llama_decode_greedy": [H, u32, u32, C.POINTER(u32)],
    "nfai_hip_llama_decode_enqueue": [H, u32],
    "nfai_hip_llama_set_token": [H, u32],
    "nfai_hip_llama_fetch_tokens": [H, u32, C.POINTER(u32)],
    "nfai_hip_llama_prefill": [H, C.POINTER(u32), u32, C.POINTER(f32)],
    "nfai_hip_llama_ingest": [H, C.POINTER(u32), u32],
    "nfai_hip_llama_stage_step": [H, u32, vp, vp, C.POINTER(f32), C.POINTER(u32)],
    "nfai_hip_llama_token_to_device": [H, vp],
    "nfai_hip_llama_token_from_device": [H, vp],
    "nfai_hip_llama_reset": [H],
    "nfai_hip_llama_set_pos": [H, u32],
    "nfai_hip_llama_pos": [H, C.POINTER(u32)],
    "nfai_hip_llama_read": [H, i32, C.POINTER(f32), u64],
    "nfai_hip_llama_read_kv": [H, u32, i32, u32, C.POINTER(f32)],
    "nfai_hip_llama_bytes_per_token": [H, u32, C.POINTER(u64), C.POINTER(u64)],
    "nfai_hip_llama_profile_step": [H, u32, C.POINTER(f32), C.POINTER(u32)],
    "nfai_hip_llama_profile_kernel": [H, u32, i32, u32, C.POINTER(f32)],
    "nfai_hip_pp_unique_id": [C.POINTER(C.c_uint8)],
    "nfai_hip_pp_init": [H, u32, u32, C.POINTER(C.c_uint8), C.POINTER(H)],
    "nfai_hip_pp_destroy": [H],
    "nfai_hip_pp_begin": [H],
    "nfai_hip_pp_end": [H],
    "nfai_hip_pp_send_hidden": [H, vp, u32, u32],
    "nfai_hip_pp_recv_hidden": [H, vp, u32, u32],
    "nfai_hip_pp_send_token": [H, vp, u32],
    "nfai_hip_pp_recv_token": [H, vp, u32],
    "nfai_hip_pp_bcast_token": [H, vp, u32],
    "nfai_hip_pp_exchange": [H, vp, u32],
    "nfai_hip_pp_info": [H, C.POINTER(u32), C.POINTER(u32), C.POINTER(i32), C.c_char_p],
    "nfai_hip_pp_check": [H],
    "nfai_hip_pp_wait": [H, u32],
    "nfai_hip_pp_abort": [H],
}

_lib = None


def load() -> C.CDLL:
    """dlopen the in-tree library; raise if it has not been built (python -m nfai_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: build it with `python -m nfai_amd.build` (hipcc, gfx950). "
                "The HIP backend has no CPU fallback.")
        L = C.CDLL(SO_PATH)
        L.nfai_hip_last_error.restype = C.c_char_p
        L.nfai_hip_last_error.argtypes = []
        L.nfai_hip_abi_version.restype = i32
        L.nfai_hip_abi_version.argtypes = []
        lax = bool(os.environ.get("NFAI_HIP_LIB_LAX"))  # A/B runs against a library built from an older revision (tools/build_ref.sh)
        for name, args in SIGNATURES.items():
            if lax and not hasattr(L, name):
                continue
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype = i32
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != OK:
        msg = load().nfai_hip_last_error().decode("utf-8", "replace")
        raise (KVCacheFull if rc == ERR_KV_FULL else NfaiHipError)(rc, msg)


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args))
