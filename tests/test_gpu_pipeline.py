"""The multi-GPU path on the one card of the GPU box.

* `HipStage` x 2 (and x 3) driven through `schedule_ticks` by ONE process with device-to-device copies as the exchange
  (`run_schedule_in_process`): stage graphs, token hand-over on the device, slots that alias one set of weights
  (`nfai_hip_llama_share_tensors`), fp16 and Q4_K_M-style weights.  Every in-flight sequence must produce exactly the tokens
  of a single-stage greedy decode (a stage is a contiguous slice of the block loop, LlamaModel.cs:118-121: bit-identical).
* `RcclComm` (nfai_hip_pp_*: RCCL bound with dlopen, operations on the stage stream) with a one-rank communicator: a grouped
  send-to-self / receive-from-self of a hidden state and of a token.  Two RCCL ranks cannot share one device, so the
  multi-rank exchange itself runs only on the driver's multi-GPU node.
"""
import numpy as np
import pytest

import oracle as orc
from nfai_amd import synth

pytestmark = pytest.mark.gpu

Q4_K, Q6_K = 12, 14


@pytest.fixture(scope="module")
def env():
    import torch
    from nfai_amd.hip import HipBufferManager
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    mgr = HipBufferManager(0, stream=stream.cuda_stream)
    yield torch, stream, mgr
    mgr.Dispose()


def device_weights(torch, dims, quant):
    """name -> (device tensor, ggml type, rows, cols) as HipStage takes them, plus the host weights the oracle takes."""
    w = synth.make_weights(dims, seed=81, std=0.05)
    dev, host = {}, {}
    for name, a in w.items():
        if a.ndim == 1:
            dev[name] = (torch.from_numpy(a).cuda(), 0, 1, a.shape[0])
            host[name] = a
        elif quant:
            qt = Q6_K if name.endswith(("attn_v.weight", "ffn_down.weight")) or name.startswith(("token_embd", "output.")) else Q4_K
            f = a.astype(np.float32)
            raw = orc.quantize_q4k(f) if qt == Q4_K else orc.quantize_q6k(f)
            deq = (orc.dequant_q4k if qt == Q4_K else orc.dequant_q6k)(raw, a.size).reshape(a.shape)
            dev[name] = (torch.from_numpy(np.ascontiguousarray(raw)).cuda(), qt, a.shape[0], a.shape[1])
            host[name] = deq
        else:
            dev[name] = (torch.from_numpy(a).cuda(), 1, a.shape[0], a.shape[1])
            host[name] = a
    return dev, host


def stage_weights(dev, dims, lb, le, first, last):
    out = {}
    for name, t in dev.items():
        if name.startswith("blk."):
            if lb <= int(name.split(".")[1]) < le:
                out[name] = t
        elif name == "token_embd.weight":
            if first or (last and dims.tied):
                out[name] = t
        elif last:
            out[name] = t
    return out


@pytest.mark.parametrize("quant", [False, True], ids=["f16", "q4_k_m"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_stages_through_schedule_in_process(env, world, quant):
    torch, stream, mgr = env
    from dataclasses import replace
    from nfai_amd.pipeline import HipStage, partition_layers, run_schedule_in_process
    dims = synth.TINY_D128  # 3 blocks, untied lm_head
    if world == 8:          # BASELINE config 5's shape of the schedule: 8 stages, 8 sequences in flight, one block per stage
        dims = replace(dims, L=8, name=dims.name + "-8blk")
    n_steps, C = (12, 32) if world < 8 else (6, 16)
    with torch.cuda.stream(stream):
        dev, host = device_weights(torch, dims, quant)
        ranges = partition_layers(dims.L, world)
        stages = [HipStage(torch, mgr, dims, (lb, le), stage_weights(dev, dims, lb, le, r == 0, r == world - 1), world, C, r, world)
                  for r, (lb, le) in enumerate(ranges)]
        first = [(11 + 5 * s) % dims.V for s in range(world)]
        run_schedule_in_process(stages, n_steps, first, lambda dst, src: dst.copy_(src))
        stream.synchronize()
        got = [stages[-1].models[s].FetchTokens(n_steps).tolist() for s in range(world)]
    for s in range(world):
        ref = orc.OracleLlama(orc.LlamaDesc(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, C=C), host)
        tok, want = first[s], []
        for _ in range(n_steps):
            tok = orc.argmax(ref.step(tok))
            want.append(tok)
        assert got[s] == want, (s, got[s], want)
    # the slots of a stage alias slot 0's weights: one copy of the stage's bytes, whatever the number of sequences in flight
    for st in stages:
        b0 = st.models[0].BytesPerToken(0)[0]
        assert all(m.BytesPerToken(0)[0] == b0 for m in st.models)
    for st in stages:
        st.dispose()


def test_two_stages_at_8b_widths(env):
    """Two stages of ONE block each at the Llama-3.1-8B widths (E 4096, 32 / 8 heads of 128, F 14336: the kernels and launch
    geometries BASELINE config 5 runs), vocabulary cut to 2048: both in-flight sequences must produce exactly the tokens of a
    single-stage greedy decode of the same two blocks on the GPU (a stage is a slice of the block loop: bit-identical)."""
    torch, stream, mgr = env
    from dataclasses import replace
    from nfai_amd.llama_model import LlamaModel
    from nfai_amd.pipeline import HipStage, run_schedule_in_process
    dims = replace(synth.LLAMA_31_8B, L=2, V=2048, name="llama-3.1-8b-2blk")
    n_steps, C, world = 8, 16, 2
    with torch.cuda.stream(stream):
        g = torch.Generator(device="cuda")
        g.manual_seed(5)
        dev = {}
        for name, shape in dims.shapes().items():
            if len(shape) == 1:
                dev[name] = (1.0 + 0.1 * torch.randn(shape, device="cuda", generator=g), 0, 1, shape[0])
            else:
                dev[name] = ((0.02 * torch.randn(shape, device="cuda", generator=g)).half(), 1, shape[0], shape[1])
        stages = [HipStage(torch, mgr, dims, (r, r + 1), stage_weights(dev, dims, r, r + 1, r == 0, r == 1), world, C, r, world) for r in range(2)]
        first = [7, 1234]
        run_schedule_in_process(stages, n_steps, first, lambda dst, src: dst.copy_(src))
        stream.synchronize()
        got = [stages[-1].models[s].FetchTokens(n_steps).tolist() for s in range(world)]
        tens = {k: (t.data_ptr(), ty, rows, cols) for k, (t, ty, rows, cols) in dev.items()}
        d = dict(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, eps=1e-5, rope_dims=dims.D, rope_base=500000.0)
        whole = LlamaModel(mgr, {"general.name": dims.name}, tens, C, dims=d)
        for s in range(world):
            assert got[s] == whole.Greedy(first[s], n_steps).tolist()
            whole.Reset()
        whole.Dispose()
        for st in stages:
            st.dispose()


def test_single_stage_schedule_world1(env):
    """world == 1 goes through the same schedule (first + last_from_first): tokens == single-model greedy."""
    torch, stream, mgr = env
    from nfai_amd.llama_model import LlamaModel
    from nfai_amd.pipeline import HipStage, run_schedule_in_process
    dims = synth.TINY
    with torch.cuda.stream(stream):
        dev, host = device_weights(torch, dims, False)
        st = HipStage(torch, mgr, dims, (0, dims.L), dev, 1, 24, 0, 1)
        run_schedule_in_process([st], 10, [9], lambda dst, src: dst.copy_(src))
        stream.synchronize()
        got = st.models[0].FetchTokens(10).tolist()
        m = LlamaModel(mgr, synth.make_metadata(dims), host, 24)
        assert got == m.Greedy(9, 10).tolist()
        m.Dispose()
        st.dispose()


def test_rccl_exchange_one_rank(env):
    """nfai_hip_pp_* on hardware: communicator of one rank, a group holding a send-to-self and the matching receive for a
    hidden state (E floats) and for a token word, enqueued on the stage stream."""
    torch, stream, mgr = env
    from nfai_amd.pipeline import RcclComm
    with torch.cuda.stream(stream):
        comm = RcclComm(mgr, 0, 1, RcclComm.unique_id())
        a = torch.arange(3072, device="cuda", dtype=torch.float32) * 0.5
        b = torch.zeros(3072, device="cuda", dtype=torch.float32)
        ta = torch.tensor([123456], device="cuda", dtype=torch.int32)
        tb = torch.zeros(1, device="cuda", dtype=torch.int32)
        stream.synchronize()
        comm.exchange([(a, 0), (ta, 0)], [(b, 0), (tb, 0)])           # one native call per tick (nfai_hip_pp_exchange)
        stream.synchronize()
        assert torch.equal(a, b) and int(tb.item()) == 123456
        b.zero_(); tb.zero_()
        comm.exchange([(a, 0), (ta, 0)], [(b, 0), (tb, 0)])           # the cached operation array of the repeating pattern
        comm.exchange_per_op([(b, 0)], [(a, 0)])                        # and the per-operation entry points
        stream.synchronize()
        assert torch.equal(a, b) and int(tb.item()) == 123456
        view = comm.info()                                              # RCCL's own record of the communicator
        assert view["nranks"] == 1 and view["rank"] == 0 and view["device"] == 0 and len(view["pci_bus_id"]) >= 7
        from nfai_amd import _lib as L
        with pytest.raises(L.NfaiHipError, match="bad operation"):
            bad = (L.PpOp * 1)(L.PpOp(a.data_ptr(), 8, 5, 0, 0))       # peer 5 of 1
            L.call("nfai_hip_pp_exchange", comm.handle, bad, 1)
        import ctypes as C
        from nfai_amd import _lib
        _lib.call("nfai_hip_pp_bcast_token", comm.handle, C.c_void_p(tb.data_ptr()), 0)
        stream.synchronize()
        assert int(tb.item()) == 123456
        # failure detection: no asynchronous error on a healthy communicator; the bounded wait returns at once on an idle stream,
        # covers enqueued work, and reports a stream that does not drain by its deadline (a long-running kernel stands in for a
        # peer that never posts its half of an exchange), naming the rank
        comm.check()
        comm.wait(5.0)
        comm.exchange([(a, 0), (ta, 0)], [(b, 0), (tb, 0)])
        comm.wait(30.0)
        assert torch.equal(a, b)
        big = torch.empty(1 << 28, device="cuda", dtype=torch.float32)
        for _ in range(40):
            big.mul_(1.0001)                                            # ~40 x 2 GB of traffic: tens of milliseconds on the stage stream
        with pytest.raises(L.NfaiHipError, match="rank 0 of 1: the stage stream did not drain within 1 ms"):
            comm.wait(0.001)
        comm.wait(60.0)
        del big
        comm.close()


def test_pp_bad_arguments(env):
    import ctypes as C
    from nfai_amd import _lib
    torch, stream, mgr = env
    h = _lib.H()
    uid = (C.c_uint8 * 128)()
    with pytest.raises(_lib.NfaiHipError, match="bad arguments"):
        _lib.call("nfai_hip_pp_init", mgr.handle, 3, 2, uid, C.byref(h))
    with pytest.raises(_lib.NfaiHipError, match="invalid pipeline handle"):
        _lib.call("nfai_hip_pp_begin", 12345)


def test_bench_two_rank_rehearsal_on_one_card():
    """`bench.py --gpus 2` started WITHOUT a launcher (the form the driver uses for N = 1) must start its own
    torch.distributed.run child; with NFAI_PP_REHEARSAL=1 both ranks share device 0 and exchange over gloo: HipStage + TorchComm
    + the wavefront schedule in two real processes.  Checks the contract fields of the one JSON line rank 0 prints."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NFAI_PP_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "1", "--model", "llama-3.2-1b",
                        "--context", "96"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["parallelism"] == "pp2" and len(d["config"]["layer_ranges"]) == 2
    # the single-stream figure beside the aggregate ("`world` sequences in flight" stated), per-phase deadlines,
    # and BASELINE config 5 (Llama-3.1-8B Q4_K_M over the same stages) as a second entry of the same line
    assert d["sequences_in_flight"] == 2 and 0 < d["single_stream_tokens_per_s"] and "2 independent" in d["config"]["workload"]
    assert d["single_stream_tokens_per_s"] <= 1.05 * d["value"]          # one stream is never faster than the full pipeline's aggregate
    assert d["config"]["phase_deadlines_s"]["first exchanges"] > 0 and "RCCL did not run" in d["config"]["multi_rank_rccl_note"]
    c5 = d["configs"][0]
    assert "config 5" in c5["baseline_config"] and "llama-3.1-8b Q4_K_M" in c5["config"]["workload"] and c5["value"] > 0
    assert c5["single_stream_tokens_per_s"] > 0 and len(c5["config"]["layer_ranges"]) == 2 and c5["config"]["layer_ranges"][-1][1] == 32
