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


def _worker(rank, world, port, n_steps, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nfai_amd.pipeline import TorchComm
    dims = synth.TINY_D128
    w = synth.make_weights(dims, seed=41, std=0.05)
    ranges = partition_layers(dims.L, world)
    stage = _OracleStage(torch, dims, w, ranges[rank], world, n_steps + 1)
    first = [3 + 11 * s for s in range(world)]
    run_schedule(stage, TorchComm(dist), rank, world, n_steps, first)
    dist.barrier()
    if rank == world - 1:
        q.put(stage.tokens)
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_pipeline_schedule_gloo(world):
    import torch.multiprocessing as mp
    import oracle as orc
    n_steps = 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    dims = synth.TINY_D128
    w = synth.make_weights(dims, seed=41, std=0.05)
    desc = orc.LlamaDesc(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, C=n_steps + 1)
    for s in range(world):
        ref = orc.OracleLlama(desc, w)
        tok, want = 3 + 11 * s, []
        for _ in range(n_steps):
            tok = orc.argmax(ref.step(tok))
            want.append(tok)
        assert got[s] == want, (s, got[s], want)


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
