#!/usr/bin/env python3
"""bench.py — decode tokens/s of the HIP Llama-3 path on MI355X, with roofline and CPU baseline.

    python bench.py --gpus 1 --steps 128 --warmup 8            (default: N=1, finishes in minutes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json metric "decode tokens/sec Llama-3.2-3B batch=1"): Llama-3.2-3B, fp16 GGUF
weights kept fp16 in HBM, fp32 activations and fp32 KV cache (the reference's precision), batch-1
greedy decode of `steps` tokens after a 512-token context (positions 512..512+steps-1), synthetic
random-init weights generated directly in HBM (no checkpoints offline).  A "step" = one token
through all 28 blocks + lm_head + argmax, token fed back on the device, one hipGraph replay.

N > 1: the blocks are sharded as a layer pipeline (one contiguous range per rank, hidden state
handed over with RCCL send/recv); N independent sequences are kept in flight so every stage is
busy; value = tokens of all sequences / time ("weak": per-GPU bytes per step are constant).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_MEASURED_CEILING_GBPS = 6290.0  # the guide's measured float4-copy ceiling (SURVEY 8d asks for the fraction of both)
CONTEXT = 512


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="llama-3.2-3b")
    ap.add_argument("--context", type=int, default=CONTEXT)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=0, help="tokens of the CPU baseline sample (0 = auto)")
    ap.add_argument("--quant", default="f16", choices=["f16", "q4_k_m"], help="GGUF file type of the synthetic weights")
    ap.add_argument("--kv-f16", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-mfma-prefill", action="store_true", help="fill the context through the decode path")
    ap.add_argument("--profile-steps", type=int, default=6)
    ap.add_argument("--engine", type=int, default=-1, help="1 / 0: one weight-streaming engine launch per block / the five-launch path (-1 = library default)")
    ap.add_argument("--stage-blocks", default="", help="B0:B1 = run only blocks [B0, B1) of the model (+ embedding + lm_head) as a one-process model: "
                                                        "the per-GPU share of a layer pipeline (BASELINE config 5); tensor types follow the global block index")
    ap.add_argument("--configs", default="auto", help="auto: the default N=1 run appends BASELINE configs 4 (3B Q4_K_M), 2 (1B fp16) and 5's per-GPU share (8B Q4_K_M, "
                                                      "4 blocks + lm_head) as `configs[]`, each measured in a fresh child process; none: only the headline workload")
    ap.add_argument("--pp-configs", default="auto", help="N > 1: auto = BASELINE config 5 (Llama-3.1-8B Q4_K_M over the same N stages) follows the headline "
                                                         "entry as configs[0]; none = only the headline")
    ap.add_argument("--child", action="store_true", help="(internal) one `configs[]` entry: compact object, bounded CPU work")
    ap.add_argument("--sample-tokens", type=int, default=64, help="tokens of the sampling-path (decode_topk + host nucleus) and blocking-greedy timings (0 = skip)")
    return ap.parse_args()


Q4_K, Q6_K = 12, 14


def use_more_bits(i: int, n: int) -> bool:
    """llama.cpp's Q4_K_M rule for which blocks keep attn_v / ffn_down in Q6_K."""
    return i < n // 8 or i >= 7 * n // 8 or (i - n // 8) % 3 == 2


def tensor_type(name: str, dims, quant: str, layer_offset: int = 0, n_layers_file: int | None = None) -> int:
    """ggml type of a matrix under the requested file type (norm gains are always F32).  A stage share (--stage-blocks) names its
    blocks blk.0.. but takes the types of blocks layer_offset.. of the n_layers_file-block file."""
    if quant == "f16":
        return 1
    assert quant == "q4_k_m"
    if name in ("token_embd.weight", "output.weight"):
        return Q6_K
    if name.startswith("blk.") and name.endswith(("attn_v.weight", "ffn_down.weight")):
        return Q6_K if use_more_bits(int(name.split(".")[1]) + layer_offset, n_layers_file or dims.L) else Q4_K
    return Q4_K


def gen_weights_hbm(torch, dims, layer_range, first, last, seed=1234, quant="f16", layer_offset=0, n_layers_file=None):
    """Random-init weights of the architecture generated directly in HBM: name -> (tensor, ggml_type,
    rows, cols).  fp16: N(0, 0.02^2) matrices, 1 + N(0, 0.1^2) gains (nfai_amd.synth distribution).
    q4_k_m: random K-quant super-blocks (random 4/6-bit codes and 6/8-bit sub-scales, fixed small
    fp16 d / dmin so the dequantised weights are ~0.02 in magnitude) in the Q4_K_M tensor mix."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    out = {}
    lb, le = layer_range
    for name, shape in dims.shapes().items():
        if name.startswith("blk."):
            l = int(name.split(".")[1])
            if not (lb <= l < le):
                continue
        elif name == "token_embd.weight":
            if not (first or (last and dims.tied)):
                continue
        elif not last:
            continue
        if len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, device="cuda", dtype=torch.float32, generator=g)
            out[name] = (t, 0, 1, shape[0])
            continue
        ty = tensor_type(name, dims, quant, layer_offset, n_layers_file)
        if ty == 1:
            t = torch.empty(shape, device="cuda", dtype=torch.float16)
            rows = max(1, (1 << 26) // shape[1])
            for r0 in range(0, shape[0], rows):
                r1 = min(shape[0], r0 + rows)
                t[r0:r1] = (0.02 * torch.randn((r1 - r0, shape[1]), device="cuda", dtype=torch.float32, generator=g)).half()
        else:
            nblk = shape[0] * shape[1] // 256
            bb = 144 if ty == Q4_K else 210
            t = torch.randint(0, 256, (nblk, bb), device="cuda", dtype=torch.uint8, generator=g)
            if ty == Q4_K:
                hdr = torch.tensor([1.0e-4, 8.0e-4], dtype=torch.float16).view(torch.uint8).to("cuda")
                t[:, 0:4] = hdr
            else:
                t[:, 208:210] = torch.tensor([2.0e-5], dtype=torch.float16).view(torch.uint8).to("cuda")
        out[name] = (t, ty, shape[0], shape[1])
    torch.cuda.synchronize()
    return out


def as_model_tensors(_lib, weights):
    return {k: (t.data_ptr(), ty, rows, cols) for k, (t, ty, rows, cols) in weights.items()}


def host_weights(weights):
    """Device tensors -> what the oracle takes: fp16 matrices stay fp16 (same operand values);
    if any matrix is K-quantised, every matrix becomes its dequantised fp32 form."""
    import oracle as orc
    any_q = any(ty in (Q4_K, Q6_K) for _, ty, _, _ in weights.values())
    host = {}
    for k, (t, ty, rows, cols) in weights.items():
        a = t.cpu().numpy()
        if ty == Q4_K:
            a = orc.dequant_q4k(a.reshape(-1), rows * cols).reshape(rows, cols)
        elif ty == Q6_K:
            a = orc.dequant_q6k(a.reshape(-1), rows * cols).reshape(rows, cols)
        elif rows > 1 and any_q:
            a = a.astype(np.float32)
        host[k] = a
    return host


def probe_reference_vulkan_path():
    """the reference itself (C#/.NET 9 + Vulkan on lavapipe) would be the preferred CPU baseline.  It needs a
    .NET SDK, glslangValidator, a Vulkan loader + lavapipe ICD + validation layer (VulkanHelper.cs:14-17) and a real GGUF; this
    probes for them on the box it runs on and says what is missing (nothing is installed or fetched)."""
    import glob
    import shutil
    missing = [t for t in ("dotnet", "glslangValidator", "vulkaninfo") if shutil.which(t) is None]
    if not any(glob.glob(p) for p in ("/usr/lib/x86_64-linux-gnu/libvulkan.so*", "/usr/lib64/libvulkan.so*", "/usr/local/lib/libvulkan.so*")):
        missing.append("libvulkan")
    if not any(glob.glob(p) for p in ("/usr/share/vulkan/icd.d/lvp_icd*.json", "/etc/vulkan/icd.d/lvp_icd*.json")):
        missing.append("lavapipe ICD")
    if not glob.glob(os.path.join(ROOT, "**", "*.gguf"), recursive=True):
        missing.append("Llama GGUF file")
    return ("available" if not missing else "reference Vulkan path unavailable on this box: missing " + ", ".join(missing))


def cpu_baseline(args, dims, weights, first_token, gpu_logits, gpu_tokens, n_tokens, one_core=True):
    """The oracle (a port of the reference path: fp32 math, reference summation order, OpenMP over
    output rows) timed on this box's host cores on a bounded sample of the SAME workload: the
    first `n_tokens` tokens of the same model from position 0 (weights identical to the GPU's).
    gpu_logits / gpu_tokens: the GPU's logits and greedy tokens of positions 0..len-1 from the same start — EVERY one of them is
    compared with the oracle's (full size, full vocabulary): the sequences coincide while the greedy tokens do."""
    import oracle as orc
    host = host_weights(weights)
    C = n_tokens + 1
    desc = orc.LlamaDesc(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, C=C)
    ref = orc.OracleLlama(desc, host)
    tok = first_token
    t0 = time.perf_counter()
    diffs, same_tok, scale = [], [], 1.0
    for i in range(n_tokens):
        lg = ref.step(tok)
        tok = orc.argmax(lg)
        if i < len(gpu_logits) and all(same_tok):     # the sequences coincide while the greedy tokens do
            diffs.append(float(np.abs(lg - gpu_logits[i]).max()))
            scale = max(scale, float(np.abs(lg).max()))
            same_tok.append(int(tok) == int(gpu_tokens[i]))
    dt = time.perf_counter() - t0
    nt = orc.num_threads()
    out = {"value": n_tokens / dt, "unit": "tokens/s", "cores": nt, "kind": "port",
           "sample": f"first {n_tokens} greedy tokens of the same model from position 0, i.e. at positions 0..{n_tokens - 1} (the GPU leg is timed "
                     f"at positions {args.context}+: the CPU port's time per token is dominated by the weights, not the context) "
                     f"(oracle/nfai_oracle.c, fp32 math, {nt} OpenMP threads)",
           "seconds": dt,
           "max_abs_logit_diff_vs_gpu_token0": diffs[0] if diffs else None,
           "positions_compared": len(diffs), "max_abs_logit_diff_vs_gpu": max(diffs) if diffs else None, "max_abs_logit": scale,
           "max_abs_logit_diff_vs_gpu_by_position": [float(f"{d:.3g}") for d in diffs], "greedy_tokens_equal_by_position": same_tok}
    if one_core:
        # the same port on ONE core (SURVEY 8d asks for both): 2 tokens from position 0
        orc.set_num_threads(1)
        ref1 = orc.OracleLlama(desc, host)
        t1 = time.perf_counter()
        tk1 = first_token
        for _ in range(min(2, n_tokens)):
            tk1 = orc.argmax(ref1.step(tk1))
        dt1 = time.perf_counter() - t1
        orc.set_num_threads(nt)
        out.update({"value_1core": min(2, n_tokens) / dt1, "sample_1core": f"first {min(2, n_tokens)} tokens, 1 thread",
                    "reference_vulkan_path": probe_reference_vulkan_path()})
    return out


_CLASS_OF = {"attn_q.weight": "qkv", "attn_k.weight": "qkv", "attn_v.weight": "qkv", "attn_output.weight": "wo",
             "ffn_gate.weight": "gateup", "ffn_up.weight": "gateup", "ffn_down.weight": "down"}
_CLASS_NAME = {"qkv": "RMSNorm + Wq|Wk|Wv GEMV + RoPE + KV-row write", "attn": "attention over the KV cache (scores, softmax, weighted V)",
               "attn+wo": "attention over the KV cache + Wo GEMV + residual (one launch)", "wo": "Wo GEMV + residual",
               "gateup": "RMSNorm + Wgate|Wup GEMV + SiLU*up", "down": "Wdown GEMV + residual",
               "lmhead": "output RMSNorm + lm_head GEMV + ArgMax (once per token)"}


def kernel_class_bytes(weights, dims, pos, kv_elem_bytes):
    """Algorithmic bytes per LAUNCH of every decode kernel class (on-disk bytes of the tensors the launch reads, averaged
    over the blocks — Q4_K_M files mix Q4_K and Q6_K per block; attention: K and V rows 0..pos of one block)."""
    per = {"qkv": 0, "wo": 0, "gateup": 0, "down": 0}
    for name, (t, ty, rows, cols) in weights.items():
        if name.startswith("blk.") and rows > 1:
            per[_CLASS_OF[name.split(".", 2)[2]]] += t.numel() * t.element_size()
    per = {k: v / dims.L for k, v in per.items()}
    per["attn"] = 2 * dims.Hkv * dims.D * (pos + 1) * kv_elem_bytes
    head = weights.get("output.weight") or weights["token_embd.weight"]
    per["lmhead"] = head[0].numel() * head[0].element_size()
    return per


def run_single(args):
    import torch
    from dataclasses import replace
    from nfai_amd import _lib, synth
    from nfai_amd.hip import HipBufferManager
    from nfai_amd.llama_model import LlamaModel, SamplingUtils

    t_start = time.perf_counter()
    dims = synth.BY_NAME[args.model]
    layer_offset, n_layers_file, stage_note = 0, None, ""
    if args.stage_blocks:
        b0, b1 = (int(v) for v in args.stage_blocks.split(":"))
        assert 0 <= b0 < b1 <= dims.L, args.stage_blocks
        layer_offset, n_layers_file = b0, dims.L
        stage_note = (f" - blocks [{b0},{b1}) of {dims.L} + token embedding + lm_head as one process: one GPU's share of the {dims.L // (b1 - b0)}-stage "
                      f"layer pipeline (the last stage also owns the lm_head; the embedding row is read on the first)")
        dims = replace(dims, L=b1 - b0)
    torch.cuda.set_device(0)
    weights = gen_weights_hbm(torch, dims, (0, dims.L), True, True, quant=args.quant, layer_offset=layer_offset, n_layers_file=n_layers_file)
    mgr = HipBufferManager(0)
    n_cmp = 0 if args.no_cpu_baseline else (args.cpu_tokens or 128)  # ~10-15 s of CPU work on the box's 16 host threads at 3B fp16
    C = max(args.context + args.warmup + args.steps, n_cmp + 1, 16 + args.warmup + args.steps)
    m = LlamaModel(mgr, synth.make_metadata(dims), as_model_tensors(_lib, weights), C,
                   graph=not args.no_graph, kv_f16=args.kv_f16, max_batch=args.context, engine=(None if args.engine < 0 else bool(args.engine)), dims=dict(
                       E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, eps=1e-5,
                       rope_dims=dims.D, rope_base=500000.0))
    first_token = 128000 % dims.V
    # positions 0..n_cmp-1 with logits, greedy: the parity side of the CPU baseline at full size — EVERY token the oracle computes
    # is compared (multi-position attention through all blocks), not only token 0
    gpu_logits, gpu_tokens, tok = [], [], first_token
    for _ in range(n_cmp):
        lg, tok = m.Step(tok)
        gpu_logits.append(lg)
        gpu_tokens.append(tok)
    m.Reset()
    # ---- context: `context` prompt tokens through the batched MFMA prefill (BASELINE config "512-token prefill +
    #      128-token decode"), timed separately; K-quant blocks are widened to fp16 copies and take the
    #      same GEMMs.  --no-mfma-prefill fills the cache through the decode path as the reference does.  Untimed for `value`.
    prefill = None
    prompt = synth.make_tokens(dims, args.context, seed=99)
    prompt[0] = first_token
    if not args.no_mfma_prefill:
        # parity side-check of the MFMA prefill at the benchmark's own size: the same prompt through the decode path token by
        # token (the reference's way, LlamaModel.cs:103-126) must give the same argmax and logits within the fp16 tolerance
        m.SetToken(int(prompt[0]))
        for t in prompt[:-1]:
            m.Step(int(t), want_logits=False)
        lg_dec, am_dec = m.Step(int(prompt[-1]))
        m.Reset()
        mgr.Synchronize()
        mgr.TimerBegin()
        lg_pf = m.Prefill(prompt)                     # also warms the workspace (first touch); K-quant models: widens every block's matrices
        pf_first = mgr.TimerEnd()                     # to fp16 ONCE (kept for later prefills when they fit a quarter of the HBM)
        pf_err = float(np.abs(lg_pf - lg_dec).max())
        pf_tol = 2e-2 * max(1.0, float(np.abs(lg_dec).max()))   # the stated fp16 tolerance (about 1e-1 absolute at 3B; 0.021 observed)
        assert int(np.argmax(lg_pf)) == am_dec and pf_err <= pf_tol, f"MFMA prefill disagrees with the decode path: {pf_err} > {pf_tol}"
        pf_runs = []
        for _ in range(3):   # three timed shots (each from an empty cache); the median is reported
            m.Reset()
            mgr.Synchronize()
            mgr.TimerBegin()
            m.Prefill(prompt, want_logits=False)
            pf_runs.append(mgr.TimerEnd())
        pf_ms = sorted(pf_runs)[1]
        T = args.context
        per_layer = 2 * T * (2 * dims.H * dims.D * dims.E + 2 * dims.Hkv * dims.D * dims.E + 3 * dims.F * dims.E)
        attn = 4 * dims.H * dims.D * T * T // 2          # causal half of QK^T and PV
        flops = dims.L * (per_layer + attn) + 2 * dims.V * dims.E
        mfma_util = None
        try:   # MFMA-busy counters of these GEMMs (separate rocprofv3 --pmc passes, tools/prefill_pmc.py), collected offline on this build
            pm_name = next(n for n in ("round4_prefill_pmc_final.json", "round4_prefill_pmc.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
            pm = json.load(open(os.path.join(ROOT, "profiles", pm_name)))
            mfma_util = {"value": pm.get("mfma_busy_frac"), "by_kernel": pm.get("mfma_busy_frac_by_kernel"),
                         "source": f"profiles/{pm_name} (tools/prefill_pmc.py: SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x duration x 2.4 GHz) per GEMM launch, "
                                   "one rocprofv3 --pmc pass per counter group); collected offline on this build, NOT in this run"}
        except (OSError, ValueError, StopIteration):
            pass
        prefill = {"tokens": T, "ms": pf_ms, "ms_runs": pf_runs, "ms_first_call": pf_first, "tokens_per_s": T / (pf_ms * 1e-3), "tflops": flops / (pf_ms * 1e-3) / 1e12,
                   "peak_tflops": 2500.0, "frac_of_mfma_peak": flops / (pf_ms * 1e-3) / 1e12 / 2500.0, "mfma_util": mfma_util,
                   "check": {"vs": "the same prompt token by token through the decode path on the GPU", "max_abs_logit_diff": pf_err,
                             "tolerance": pf_tol, "same_argmax": True},
                   "kernel": "k_gemm_f16_glds (mfma_f32_16x16x32_f16, direct-to-LDS staging, eight waves per workgroup: 256x128 tiles on gate|up, "
                             "128x{48,64,80,96} tiles with two wave groups splitting the k-steps elsewhere; SiLU*up, residual and RoPE + q / KV-cache "
                             "stores fused into the epilogues) + k_attn_prefill (causal attention of the chunk in one launch)"
                             + ("" if args.quant == "f16" else "; K-quant matrices widened to fp16 copies by the FIRST prefill call (`ms_first_call`: also "
                                "first-touch of the workspace, logits copy) and kept while they fit a quarter of the HBM - the timed calls reuse them; "
                                "NFAI_PREFILL_WIDE_ALL=0: one block's scratch, widened per block and chunk")}
    else:
        m.SetToken(first_token)
        m.Enqueue(args.context)
        mgr.Synchronize()
    # ---- warmup, then the timed region: exactly `steps` graph replays, events on the launch stream
    m.Enqueue(args.warmup)
    mgr.Synchronize()
    torch.cuda.synchronize()
    pos0 = m.Pos
    t0 = time.perf_counter()
    mgr.TimerBegin()
    m.Enqueue(args.steps)
    ev_ms = mgr.TimerEnd()
    mgr.Synchronize()
    wall = time.perf_counter() - t0
    toks = m.FetchTokens(args.steps)
    ms_per_step = ev_ms / args.steps
    value = args.steps / (ev_ms / 1e3)
    # ---- roofline: dominant kernel = the fused RMSNorm + Wgate/Wup GEMV + SiLU*up launch
    pos_mid = pos0 + args.steps // 2
    b_tok, dom_bytes = m.BytesPerToken(pos_mid)
    prof = {}
    for _ in range(args.profile_steps):
        if m.Pos >= C:
            m.SetPos(pos0)
        for k, (ms, n) in m.ProfileStep(int(toks[-1])).items():
            a = prof.setdefault(k, [0.0, 0])
            a[0] += ms
            a[1] += n
    # the dominant kernel's duration: its launches of one step (one per block, each streaming its own weights) replayed back
    # to back, 4 rounds, inside ONE event pair on the launch stream — what rocprofv3 --kernel-trace reports per launch (the
    # per-launch event pairs above add ~2.5 us of launch overhead each)
    if m.Pos >= C:
        m.SetPos(pos0)
    engine_on = bool(prof) and prof.get("engine", (0.0, 0))[1] > 0
    if engine_on:
        # one engine launch per block: Wo + gate|up + Wdown of the block and q|k|v of the next one (the last block has none)
        bpw = 2 if args.quant == "f16" else None
        HD, KD = dims.H * dims.D, dims.Hkv * dims.D
        dom_bytes = (dims.E * HD + 3 * dims.F * dims.E) * bpw + (dims.L - 1) * (HD + 2 * KD) * dims.E * bpw // dims.L
        dom_cls, dom_name = "engine", "k_engine (Wo + residual -> RMSNorm + gate|up + SiLU*up -> Wdown + residual -> RMSNorm + next q|k|v + RoPE, one launch per block)"
        pmc_key = "nfai::k_engine"
    else:
        dom_cls = "gateup"
        dom_name = ("k_gemv<F16,GATEUP>" if args.quant == "f16" else "k_gemv_kqt<Q4_K_T16,GATEUP> int8-MFMA") + " (RMSNorm + Wgate/Wup GEMV + SiLU*up)"
    # HBM traffic of the dominant kernel from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes, gfx950
    # correction applied): collected offline on the same build by tools/pmc_traffic.py, committed under profiles/
    traffic, traffic_source = None, None
    if args.model == "llama-3.2-3b" and not args.stage_blocks:
        tag = "" if args.quant == "f16" else "_q4km"
        for rnd in ("round4", "round3", "round2", "round1"):  # newest committed PMC summary of this workload
            f = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic{tag}.json")
            try:
                pmc = json.load(open(f))
                if bool(pmc.get("engine", False)) != engine_on:
                    continue  # counters of the other launch structure
                k = pmc.get("dominant_kernel") or ("nfai::k_gemv<1, 3, 2, 3, false, true>" if args.quant == "f16" else "nfai::k_gemv_kqt<112, 3, 1, true, 0>")
                traffic = pmc["kernels"][k]["hbm_bytes_per_launch"]
                traffic_source = (f"profiles/{os.path.basename(f)}: separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over `{pmc.get('command', '?')}`, "
                                  "(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch; collected offline on this build, NOT in this run")
                break
            except (OSError, KeyError, ValueError):
                continue
    gu_eager_ms = prof[dom_cls][0] / max(1, prof[dom_cls][1]) if prof else float("nan")
    gu_ms = m.ProfileKernel(int(toks[-1]), dom_cls, 4) * 1e-3
    achieved = dom_bytes / (gu_ms * 1e-3) / 1e9
    per_kernel_us = {k: round(1e3 * v[0] / v[1], 3) for k, v in prof.items() if v[1]}
    # every decode kernel class against the roofline, from the same live replay (the dominant kernel is a third of the token)
    kernels = []
    if prof and not engine_on:
        cb = kernel_class_bytes(weights, dims, pos_mid, 2 if args.kv_f16 else 4)
        fused_attn_wo = prof.get("wo", (0.0, 0))[1] == 0
        for cls in ("qkv", "attn", "wo", "gateup", "down", "lmhead"):
            if prof.get(cls, (0.0, 0))[1] == 0:
                continue
            if m.Pos >= C:
                m.SetPos(pos0)
            us = m.ProfileKernel(int(toks[-1]), cls, 4)
            nbytes = cb[cls] + (cb["wo"] if cls == "attn" and fused_attn_wo else 0)
            label = "attn+wo" if cls == "attn" and fused_attn_wo else cls
            n_launch = prof[cls][1] // max(1, args.profile_steps)
            kernels.append({"class": label, "what": _CLASS_NAME[label], "launches_per_token": n_launch, "bytes_per_launch": int(nbytes),
                            "us_per_launch": round(us, 3), "gbps": round(nbytes / us / 1e3, 1), "frac": round(nbytes / us / 1e3 / HBM_PEAK_GBPS, 4),
                            "frac_of_measured_ceiling": round(nbytes / us / 1e3 / HBM_MEASURED_CEILING_GBPS, 4)})
    workload = (f"{dims.name} {'fp16-GGUF weights (fp16 in HBM)' if args.quant == 'f16' else 'Q4_K_M-GGUF weights (native K-quant blocks in HBM)'}, fp32 activations + {'fp16' if args.kv_f16 else 'fp32'} KV, "
                f"batch-1 greedy decode of {args.steps} tokens after a {args.context}-token context" + stage_note)
    out = {
        "metric": "decode tokens/sec Llama-3.2-3B batch=1; achieved HBM GB/s vs roofline",
        "value": value, "unit": "tokens/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload,
                   "positions": [pos0, pos0 + args.steps - 1], "kv_capacity": C, "parallelism": "single",
                   "graph": not args.no_graph, "launches_per_block": 2 if engine_on else (4 if args.quant == "f16" else 5),
                   "launches_per_token": (sum(v[1] for v in prof.values()) // max(1, args.profile_steps)) if prof else None,
                   "xcd_row_shares": dict(zip(("shares", "probe_us_equal_shares"), mgr.XcdShares()))},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "frac_of_measured_ceiling": achieved / HBM_MEASURED_CEILING_GBPS,
                     "measured_ceiling": HBM_MEASURED_CEILING_GBPS,
                     "measured_ceiling_note": "the MI355X guide's float4-copy figure (SURVEY 8d asks for the fraction of both); a long nt register stream "
                                              "exceeds it here: the lm_head launch's fraction of it is above 1, and profiles/round2_ldsdma_bench.txt has 7.0 TB/s",
                     "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": dom_name,
                     "bytes_per_launch": dom_bytes, "us_per_launch": gu_ms * 1e3, "us_per_launch_eager_event_pair": gu_eager_ms * 1e3,
                     "timing": "hipEvents on the launch stream around 4 rounds of the kernel's launches of one step (one per block), back to back",
                     "kernels": kernels},
        "token_hbm_gbps": b_tok / (ms_per_step * 1e-3) / 1e9,
        "token_hbm_frac_of_peak": b_tok / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "token_hbm_frac_of_measured_ceiling": b_tok / (ms_per_step * 1e-3) / 1e9 / HBM_MEASURED_CEILING_GBPS,
        "bytes_per_token": b_tok,
        "kernel_us_eager_events": per_kernel_us,
        "host_wall_ms_per_step": 1e3 * wall / args.steps,
        "prefill": prefill,
    }
    # ---- the same decode at a SHORT context (positions 16..): the like-for-like figure for per-layer comparisons (MI355X guide: 1B layer)
    m.Reset()
    m.SetToken(first_token)
    m.Enqueue(16 + args.warmup)
    mgr.Synchronize()
    sp0 = m.Pos
    mgr.TimerBegin()
    m.Enqueue(args.steps)
    s_ms = mgr.TimerEnd()
    b_short, _ = m.BytesPerToken(sp0 + args.steps // 2)
    out["short_context"] = {"positions": [sp0, sp0 + args.steps - 1], "tokens_per_s": args.steps / (s_ms * 1e-3), "ms_per_step": s_ms / args.steps,
                            "token_hbm_frac_of_peak": b_short / (s_ms / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    # ---- the loop a host actually runs (LlamaModel.cs:130,165): one BLOCKING call per token.  Greedy = decode_step with the argmax
    #      read back; the reference's default sampler = decode_topk (candidates on the device, 8k + 8 bytes back) + the host half of
    #      SamplingUtils.TopP.  `value` above is the device-side greedy loop (tokens fed back on the device, no host in the loop).
    if args.sample_tokens > 0:
        n = min(args.sample_tokens, C - 1)
        start = max(0, min(pos0, C - n))
        rng = np.random.default_rng(7)

        def host_loop(step):
            """n blocking calls from position `start`; (tokens/s over the whole loop, tokens/s at the median per-token time)."""
            m.SetPos(start)
            tk = step(int(toks[-1]))   # untimed: captures the call's graph, allocates the top-k workspace
            m.SetPos(start)
            mgr.Synchronize()
            per = []
            for _ in range(n):
                t1 = time.perf_counter()
                tk = step(tk)
                per.append(time.perf_counter() - t1)
            return n / sum(per), 1.0 / float(np.median(per))

        def sampled(tk):
            ids, probs = m.StepTopK(tk)
            return SamplingUtils.TopPFromCandidates(ids, probs, rng=rng)

        g_all, g_med = host_loop(lambda tk: m.Step(tk, want_logits=False)[1])
        k_all, k_med = host_loop(lambda tk: int(m.StepTopK(tk)[0][0]))
        s_all, s_med = host_loop(sampled)
        out["sampling_path"] = {
            "tokens": n, "positions": [start, start + n - 1],
            "blocking_greedy_tokens_per_s": g_med, "decode_topk_tokens_per_s": k_med, "sampling_path_tokens_per_s": s_med,
            "sampling_vs_blocking_greedy": s_med / g_med,
            "whole_loop_tokens_per_s": {"blocking_greedy": g_all, "decode_topk": k_all, "sampling_path": s_all},
            "what": "one blocking C-ABI call per token from this Python host, tokens/s at the MEDIAN per-token wall time (whole-loop figures beside "
                    "them): nfai_hip_llama_decode_step (argmax read back) / nfai_hip_llama_decode_topk (token kernels + the two top-40 launches + "
                    "ONE 528-byte read-back in one hipGraph, one synchronisation) / the same + SamplingUtils.TopPFromCandidates (nucleus 0.95 + "
                    "draw; the reference's default loop, LlamaModel.cs:130,165)"}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, dims, weights, first_token, gpu_logits, gpu_tokens, n_cmp, one_core=not args.child)
    else:
        out["cpu_baseline"] = None
    out["wall_s"] = time.perf_counter() - t_start
    m.Dispose()
    return out, weights


# BASELINE.json configs beside the headline (config 3), each measured in its own child process: (label, bench.py arguments)
CHILD_CONFIGS = [
    ("BASELINE config 4: Llama-3.2-3B Q4_K_M GGUF, batch=1 decode on 1xMI355X", ["--model", "llama-3.2-3b", "--quant", "q4_k_m", "--cpu-tokens", "4"]),
    ("BASELINE config 2: Llama-3.2-1B fp16 GGUF, batch=1 autoregressive decode on 1xMI355X", ["--model", "llama-3.2-1b", "--quant", "f16", "--cpu-tokens", "16"]),
    ("BASELINE config 5, one GPU's share: Llama-3.1-8B Q4_K_M, 4 of 32 blocks (the last four: attn_v / ffn_down in Q6_K) + untied Q6_K output.weight, V = 128256",
     ["--model", "llama-3.1-8b", "--quant", "q4_k_m", "--stage-blocks", "28:32", "--cpu-tokens", "8"]),
]


def run_child_configs(args):
    """Each entry in a FRESH child process started by this one (a child, never a re-exec; this process's own measurements are done and
    its model is disposed), 64 timed steps after a 512-token MFMA prefill, oracle logits at the first positions."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    entries = []
    for label, extra in CHILD_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--configs", "none", "--gpus", "1", "--steps", "64", "--warmup", "8",
               "--context", str(args.context), "--sample-tokens", "0", "--profile-steps", "3"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
            lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                entries.append({"baseline_config": label, "error": f"child exited with {r.returncode}: {r.stderr.strip()[-400:]}"})
                continue
            d = json.loads(lines[-1])
        except (subprocess.TimeoutExpired, ValueError) as e:
            entries.append({"baseline_config": label, "error": repr(e)[:400]})
            continue
        cb, rf = d.get("cpu_baseline") or {}, d["roofline"]
        entries.append({
            "baseline_config": label, "workload": d["config"]["workload"], "command": "python bench.py " + " ".join(cmd[3:]),
            "value": d["value"], "unit": "tokens/s", "ms_per_step": d["ms_per_step"], "steps": d["steps"], "positions": d["config"]["positions"],
            "launches_per_token": d["config"]["launches_per_token"],
            "bytes_per_token": d["bytes_per_token"], "token_hbm_gbps": d["token_hbm_gbps"], "token_hbm_frac_of_peak": d["token_hbm_frac_of_peak"],
            "token_hbm_frac_of_measured_ceiling": d["token_hbm_frac_of_measured_ceiling"],
            "roofline": {k: rf[k] for k in ("kernel", "bytes_per_launch", "us_per_launch", "achieved", "frac", "frac_of_measured_ceiling", "traffic", "kernels")},
            "short_context": d.get("short_context"),
            "prefill": {k: d["prefill"][k] for k in ("tokens", "ms", "tflops", "frac_of_mfma_peak", "check")} if d.get("prefill") else None,
            "parity_vs_oracle": {"positions_compared": cb.get("positions_compared"), "max_abs_logit_diff": cb.get("max_abs_logit_diff_vs_gpu"),
                                 "max_abs_logit": cb.get("max_abs_logit"), "by_position": cb.get("max_abs_logit_diff_vs_gpu_by_position"),
                                 "greedy_tokens_equal_by_position": cb.get("greedy_tokens_equal_by_position"),
                                 "oracle_tokens_per_s": cb.get("value"), "cores": cb.get("cores")},
            "child_wall_s": round(time.perf_counter() - t0, 1)})
    return entries


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU as a CHILD (never exec after the GPU was touched:
        # nothing above initialises HIP) and leave with its exit code
        import subprocess
        port = os.environ.get("MASTER_PORT", "29517")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)
    if args.gpus > 1 or world > 1:
        from nfai_amd.pipeline import run_bench_pipeline
        run_bench_pipeline(args)
    else:
        out, weights = run_single(args)
        del weights
        want = args.configs == "auto" and not args.child and args.model == "llama-3.2-3b" and args.quant == "f16" and not args.stage_blocks
        if want or args.configs == "all":
            import gc
            import torch
            gc.collect()
            torch.cuda.empty_cache()   # this process keeps the device open, but none of its 6.4 GB of weights
            out["configs"] = run_child_configs(args)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
