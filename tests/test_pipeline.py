"""The multi-GPU path on CPU: the layer-pipeline schedule of nfai_amd.pipeline.run_schedule with
world_size 2 and 3 over gloo, each stage computed by the CPU oracle.  Every in-flight sequence
must produce exactly the tokens a single-process greedy decode produces (bit-identical hidden
states: a stage is a contiguous slice of the block loop).  Also the stage partitioner."""
import os
import socket

import numpy as np
import pytest

from nfai_amd import synth
from nfai_amd.pipeline import partition_layers, run_schedule


def test_partition_layers():
    assert partition_layers(28, 1) == [(0, 28)]
    for world in (2, 4, 8):
        r = partition_layers(28, world, 229.0, 788.0)
        assert r[0][0] == 0 and r[-1][1] == 28 and len(r) == world
        assert all(a[1] == b[0] for a, b in zip(r, r[1:])) and all(e > b for b, e in r)
        cost = [(e - b) * 229.0 + (788.0 if i == world - 1 else 0.0) for i, (b, e) in enumerate(r)]
        assert max(cost) <= 1.35 * (28 * 229.0 + 788.0) / world  # balanced by streamed bytes, lm_head included
    # lm_head worth 3.4 blocks: the last stage keeps one block, the other 15 are spread evenly (the largest stage, 1 + 3.4,
    # cannot be undercut: every stage owns at least one block)
    assert partition_layers(16, 8, 1.0, 3.4) == [(0, 3), (3, 5), (5, 7), (7, 9), (9, 11), (11, 13), (13, 15), (15, 16)]
    assert partition_layers(16, 8, 1.0, 0.0) == [(b, b + 2) for b in range(0, 16, 2)]
    assert partition_layers(32, 8, 218.0, 1050.0) == [(0, 5), (5, 10), (10, 15), (15, 19), (19, 23), (23, 27), (27, 31), (31, 32)]
    assert partition_layers(3, 3) == [(0, 1), (1, 2), (2, 3)]


class _OracleStage:
    def __init__(self, torch, dims, weights, lrange, n_slots, C):
        import oracle as orc
        self.torch, self.orc, self.dims, self.lrange, self.w = torch, orc, dims, lrange, weights
        desc = orc.LlamaDesc(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, C=C)
        self.models = [orc.OracleLlama(desc, weights) for _ in range(n_slots)]
        self._hin = [torch.zeros(dims.E) for _ in range(n_slots)]
        self._hout = [torch.zeros(dims.E) for _ in range(n_slots)]
        self._tok = [torch.zeros(1, dtype=torch.int32) for _ in range(n_slots)]
        self.tokens = [[] for _ in range(n_slots)]

    def h_in(self, s):
        return self._hin[s]

    def h_out(self, s):
        return self._hout[s]

    def tok(self, s):
        return self._tok[s]

    def _run(self, slot, hidden):
        m = self.models[slot]
        h = m.layers(hidden, *self.lrange)
        m.advance()
        return h

    def first(self, slot, token):
        if token is None:
            token = int(self._tok[slot].item())
        h = self.w["token_embd.weight"][token].astype(np.float32)
        self._hout[slot].copy_(self.torch.from_numpy(self._run(slot, h)))

    def middle(self, slot):
        self._hout[slot].copy_(self.torch.from_numpy(self._run(slot, self._hin[slot].numpy())))

    def last_from_first(self, slot):
        raise AssertionError("world == 1 is the single-GPU path")

    def last(self, slot):
        orc = self.orc
        h = self._run(slot, self._hin[slot].numpy())
        xn = orc.rmsnorm(h, self.w["output_norm.weight"], 1e-5)
        head = self.w.get("output.weight", self.w["token_embd.weight"])
        t = orc.argmax(orc.gemv_f16w(head, xn))
        self.tokens[slot].append(t)
        self._tok[slot][0] = t


class _CountingComm:
    """TorchComm + the optional `check` hook run_schedule polls once per batch of `world` ticks (RCCL: ncclCommGetAsyncError)."""

    def __init__(self, inner):
        self.inner, self.checks, self.exchanges = inner, 0, 0

    def exchange(self, sends, recvs):
        self.exchanges += 1
        self.inner.exchange(sends, recvs)

    def check(self):
        self.checks += 1


def _worker(rank, world, port, n_steps, q, n_single=0):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nfai_amd.pipeline import TorchComm
    dims = synth.TINY_D128
    w = synth.make_weights(dims, seed=41, std=0.05)
    ranges = partition_layers(dims.L, world)
    stage = _OracleStage(torch, dims, w, ranges[rank], world, n_steps + n_single + 1)
    first = [3 + 11 * s for s in range(world)]
    comm = _CountingComm(TorchComm(dist))
    run_schedule(stage, comm, rank, world, n_steps, first)
    n_ticks = n_steps * world + world - 1
    assert comm.checks == n_ticks // world, (comm.checks, n_ticks)
    if n_single:
        # bench.py's second timed region: ONE sequence (slot 0) continues alone through the stages; its first token is handed
        # over by the host again, as run_bench_pipeline does
        dist.barrier()
        nxt = [stage.tokens[0][-1]] if rank == world - 1 else [None]
        dist.broadcast_object_list(nxt, src=world - 1)
        before = comm.exchanges
        run_schedule(stage, comm, rank, world, n_single, [nxt[0]] + first[1:], n_slots=1)
        # one hop per tick and link: a middle stage posts a receive and a send per token, the ends one fewer at the edges
        assert comm.exchanges - before <= 2 * n_single + 1
    dist.barrier()
    if rank == world - 1:
        q.put(stage.tokens)
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n_single", [(2, 0), (3, 0), (2, 4), (3, 3)])
def test_pipeline_schedule_gloo(world, n_single):
    """n_single > 0: after the pipeline-full phase slot 0 continues ALONE for n_single tokens (schedule_ticks(..., n_slots=1):
    bench.py's single-stream region) — its tokens must continue the single-process greedy decode exactly."""
    import torch.multiprocessing as mp
    import oracle as orc
    n_steps = 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_steps, q, n_single)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    dims = synth.TINY_D128
    w = synth.make_weights(dims, seed=41, std=0.05)
    desc = orc.LlamaDesc(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, C=n_steps + n_single + 1)
    for s in range(world):
        ref = orc.OracleLlama(desc, w)
        tok, want = 3 + 11 * s, []
        for _ in range(n_steps + (n_single if s == 0 else 0)):
            tok = orc.argmax(ref.step(tok))
            want.append(tok)
        assert got[s] == want, (s, got[s], want)


def test_phase_watchdog_and_costs():
    """The per-phase deadline of a pipeline run: a phase that finishes in time disarms it; one that overruns
    makes the rank print rank / stage / phase and leave through the exit hook (os._exit(3) in bench.py; a recorder here).
    pipeline_costs: the bytes partition_layers balances follow the file type."""
    import time
    from nfai_amd.pipeline import PhaseWatchdog, pipeline_costs
    codes = []
    wd = PhaseWatchdog(5, lambda: "stage blocks [4,8) of test-model", exit_fn=codes.append)
    with wd.phase("quick", 0.5):
        time.sleep(0.05)
    time.sleep(0.7)
    assert codes == [] and wd.fired is None           # disarmed on exit
    with wd.phase("stuck exchange", 0.15):
        time.sleep(0.6)
    assert codes == [3] and wd.fired == "stuck exchange"
    lb, hb = pipeline_costs(synth.LLAMA_31_8B, "f16")
    assert lb == (2 * 4096 * 4096 + 2 * 1024 * 4096 + 3 * 14336 * 4096) * 2 and hb == 128256 * 4096 * 2
    lq, hq = pipeline_costs(synth.LLAMA_31_8B, "q4_k_m")
    assert 0.28 * lb < lq < 0.33 * lb and abs(hq - 128256 * 4096 * 6.5625 / 8) < 1      # Q4_K with Q6_K v / down on half the blocks; Q6_K head
    r = partition_layers(32, 8, lq, hq)
    assert r[-1][1] - r[-1][0] < r[0][1] - r[0][0]    # the last stage also streams the lm_head: fewer blocks


# ---- properties of the host logic (hypothesis): any world size, any model depth ------------------------------------------------
from hypothesis import given, settings, strategies as st  # noqa: E402


@settings(max_examples=200, deadline=None)
@given(n_layers=st.integers(1, 80), world=st.integers(1, 8), layer_cost=st.floats(0.5, 500.0), head_ratio=st.floats(0.0, 12.0))
def test_partition_layers_properties(n_layers, world, layer_cost, head_ratio):
    """Contiguous, covering, every stage >= 1 block, and minimax: no other contiguous partition has a smaller largest stage
    (checked against the exact optimum, which for contiguous equal-cost blocks is a one-dimensional search over the last
    stage's size)."""
    if world > n_layers:
        world = n_layers
    head = head_ratio * layer_cost
    r = partition_layers(n_layers, world, layer_cost, head)
    assert len(r) == world and r[0][0] == 0 and r[-1][1] == n_layers
    assert all(a[1] == b[0] for a, b in zip(r, r[1:])) and all(e > b for b, e in r)
    cost = max((e - b) * layer_cost + (head if i == world - 1 else 0.0) for i, (b, e) in enumerate(r))
    if world == 1:
        best = n_layers * layer_cost + head
    else:
        best = min(max(k * layer_cost + head, -(-(n_layers - k) // (world - 1)) * layer_cost)
                   for k in range(1, n_layers - (world - 1) + 1))
    assert cost <= best * (1 + 1e-9) + 1e-9, (r, cost, best)


class _TraceStage:
    """Records what the schedule asks of a stage; buffers are (kind, slot) tokens so that sends and receives can be matched."""

    def __init__(self):
        self.calls = []

    def h_in(self, s):
        return ("h_in", s)

    def h_out(self, s):
        return ("h_out", s)

    def tok(self, s):
        return ("tok", s)

    def first(self, slot, token):
        self.calls.append(("first", slot, token))

    def middle(self, slot):
        self.calls.append(("middle", slot))

    def last(self, slot):
        self.calls.append(("last", slot))

    def last_from_first(self, slot):
        self.calls.append(("last_from_first", slot))


@settings(max_examples=40, deadline=None)
@given(world=st.integers(2, 8), n_steps=st.integers(1, 6), data=st.data())
def test_schedule_ticks_fewer_slots_properties(world, n_steps, data):
    """n_slots < world (n_slots = 1: the single-stream region of bench.py): same ticks and links, every posted send still has its
    receive in the same tick, only the jobs of slots < n_slots run, each exactly n_steps times per stage and in order."""
    from nfai_amd.pipeline import schedule_ticks
    n_slots = data.draw(st.integers(1, world))
    stages = [_TraceStage() for _ in range(world)]
    firsts = [100 + s for s in range(world)]
    gens = [schedule_ticks(stages[r], r, world, n_steps, firsts, n_slots) for r in range(world)]
    n_sends = 0
    for posted in zip(*gens):
        for r, (sends, recvs) in enumerate(posted):
            for buf, dst in sends:
                match = [b for b, src in posted[dst][1] if src == r]
                assert len(match) == 1 and match[0][1] == buf[1] < n_slots
                n_sends += 1
        assert sum(len(s_) for s_, _ in posted) == sum(len(r_) for _, r_ in posted)
    assert n_sends == n_slots * (n_steps * (world - 1) + (n_steps - 1))   # hidden hops + token returns
    for r, stg in enumerate(stages):
        kind = "first" if r == 0 else ("last" if r == world - 1 else "middle")
        jobs = [c for c in stg.calls if c[0] == kind]
        assert len(jobs) == n_steps * n_slots and {c[1] for c in jobs} == set(range(n_slots))


@settings(max_examples=60, deadline=None)
@given(world=st.integers(1, 8), n_steps=st.integers(1, 9))
def test_schedule_ticks_properties(world, n_steps):
    """For every world size and step count: in every tick each posted send has exactly one matching receive on its peer in the
    SAME tick (no deadlock under rendezvous semantics), every stage runs every (slot, step) job exactly once and in step order
    per slot, the first stage gets a host token only at step 0, and a hidden state is always consumed in the tick after it was
    produced."""
    from nfai_amd.pipeline import schedule_ticks
    stages = [_TraceStage() for _ in range(world)]
    firsts = [100 + s for s in range(world)]
    gens = [schedule_ticks(stages[r], r, world, n_steps, firsts) for r in range(world)]
    n_ticks = 0
    for posted in zip(*gens):
        n_ticks += 1
        for r, (sends, recvs) in enumerate(posted):
            for buf, dst in sends:
                match = [b for b, src in posted[dst][1] if src == r]
                assert len(match) == 1, (world, n_steps, r, dst)
                # a hidden state goes into the same slot's input; a token into the same slot's token word
                assert match[0][1] == buf[1] and {buf[0], match[0][0]} in ({"h_out", "h_in"}, {"tok"})
            for buf, src in recvs:
                assert sum(1 for b, dst in posted[src][0] if dst == r) >= 1
        assert sum(len(s_) for s_, _ in posted) == sum(len(r_) for _, r_ in posted)
    assert n_ticks == n_steps * world + world - 1
    for r, stg in enumerate(stages):
        kind = "first" if r == 0 else ("last" if r == world - 1 else "middle")
        jobs = [c for c in stg.calls if c[0] == kind]
        assert len(jobs) == n_steps * world
        per_slot = {}
        for c in jobs:
            per_slot.setdefault(c[1], []).append(c)
        assert sorted(per_slot) == list(range(world)) and all(len(v) == n_steps for v in per_slot.values())
        if r == 0:
            for slot, v in per_slot.items():
                assert v[0][2] == firsts[slot] and all(c[2] is None for c in v[1:])
    if world == 1:
        assert sum(1 for c in stages[0].calls if c[0] == "last_from_first") == n_steps


def test_bench_helpers_stage_share_types_and_child_commands():
    """bench.py's host logic that needs no GPU: a stage share (--stage-blocks) names its blocks blk.0.. but takes the tensor types of its
    GLOBAL block indices (llama.cpp's Q4_K_M "more bits" rule); every configs[] child command parses with bench.py's own parser."""
    import sys
    import bench as B
    dims = synth.LLAMA_31_8B
    for l in range(dims.L):
        want = B.Q6_K if B.use_more_bits(l, dims.L) else B.Q4_K
        assert B.tensor_type(f"blk.{l}.ffn_down.weight", dims, "q4_k_m") == want
        for b0 in (0, 8, 28):
            if b0 <= l < b0 + 4:
                assert B.tensor_type(f"blk.{l - b0}.ffn_down.weight", dims, "q4_k_m", b0, dims.L) == want
    assert all(B.tensor_type(f"blk.{i}.attn_v.weight", dims, "q4_k_m", 28, 32) == B.Q6_K for i in range(4))   # the last four blocks
    assert B.tensor_type("output.weight", dims, "q4_k_m") == B.Q6_K and B.tensor_type("blk.3.attn_q.weight", dims, "q4_k_m") == B.Q4_K
    assert B.tensor_type("blk.3.attn_q.weight", dims, "f16") == 1
    argv = sys.argv
    try:
        for label, extra in B.CHILD_CONFIGS:
            sys.argv = ["bench.py", "--child", "--configs", "none", "--gpus", "1", "--steps", "64", "--warmup", "8", "--context", "512",
                        "--sample-tokens", "0", "--profile-steps", "3"] + extra
            a = B.parse()
            assert a.child and a.configs == "none" and a.model in synth.BY_NAME and a.quant in ("f16", "q4_k_m"), label
            if a.stage_blocks:
                b0, b1 = (int(v) for v in a.stage_blocks.split(":"))
                assert 0 <= b0 < b1 <= synth.BY_NAME[a.model].L
    finally:
        sys.argv = argv
