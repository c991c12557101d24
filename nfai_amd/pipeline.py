"""Layer pipeline across the GPUs of one node: rank r owns a contiguous range of
TransformerBlocks (the block loop of LlamaModel.cs:118-121 cut into stages), its slice of the KV
cache and only those weights.  One exchange per stage boundary per token: the hidden state
(n_embd fp32 = 8-16 KB) with RCCL send/recv; the last stage returns a 4-byte token id to stage 0.
There is no collective on the data path.

Batch-1 pipeline parallelism does not speed up ONE stream (the hops only add latency), so
`world` independent sequences are kept in flight: at any time every stage works on a different
sequence.  Per-GPU bytes per step are constant in `world` ("weak" scaling): L/world blocks x world
sequences.

`run_schedule` is backend-agnostic (a stage object + a comm object) so the schedule is tested on
CPU with gloo (tests/test_pipeline.py); `HipStage`/`TorchComm` bind it to libnfai_hip.so and
torch.distributed (backend "nccl" = RCCL over xGMI).
"""
from __future__ import annotations

import json
import os
import sys
import threading
import time

import numpy as np


def partition_layers(n_layers: int, world: int, layer_cost: float = 1.0, head_cost: float = 0.0) -> list[tuple[int, int]]:
    """Contiguous block ranges that minimise the largest per-stage cost (streamed bytes); the last stage also runs
    lm_head (`head_cost` in the units of `layer_cost`), so it gets fewer blocks.  Every stage gets >= 1 block.
    Among the minimax partitions: the last stage as large as fits, the rest as even as possible (deterministic)."""
    assert 1 <= world <= n_layers

    def fits(cap: float):
        """Greedy from the back: the last stage takes as many blocks as fit beside lm_head, then each earlier one."""
        sizes, left = [], n_layers
        for r in range(world - 1, -1, -1):
            extra = head_cost if r == world - 1 else 0.0
            if r == 0:
                k = left
            else:
                k = min(int((cap - extra) / layer_cost + 1e-9), left - r)  # leave >= 1 block for each earlier stage
            if k < 1 or k * layer_cost + extra > cap * (1 + 1e-12) + 1e-9:
                return None
            sizes.append(k)
            left -= k
        return sizes[::-1] if left == 0 else None

    # candidate caps: k blocks, or k blocks + head
    cands = sorted({k * layer_cost for k in range(1, n_layers + 1)} | {k * layer_cost + head_cost for k in range(1, n_layers + 1)})
    for cap in cands:
        sizes = fits(cap)
        if sizes:
            break
    else:  # cannot happen: the largest candidate fits everything
        raise AssertionError("partition_layers: no feasible partition")
    # the last stage keeps what the back-greedy pass gave it (the most that fits beside lm_head); the other blocks are
    # spread as evenly as possible over the earlier stages (sizes differ by at most one, larger ones first)
    if world > 1:
        k_last = sizes[-1]
        rem, n = n_layers - k_last, world - 1
        sizes = [rem // n + (1 if r < rem % n else 0) for r in range(n)] + [k_last]
    bounds, start = [], 0
    for k in sizes:
        bounds.append((start, start + k))
        start += k
    return bounds


def schedule_ticks(stage, rank: int, world: int, n_steps: int, first_tokens, n_slots: int | None = None):
    """`n_steps` tokens for each of `n_slots` (default `world`: the pipeline exactly full) in-flight sequences (slots), as a wavefront: at tick t
    stage r runs job j = t - r (slot j % world, step j // world); after the compute of a tick every
    rank posts ONE batch of point-to-point operations (send my result on, receive what I need for the
    next tick).  Both ends of every link post in the same tick, so the schedule cannot deadlock even
    when a send only completes against its matching receive (RCCL/NCCL and gloo semantics).

    The token of job j (last stage, tick j + world - 1) is the input of job j + world (stage 0, tick
    j + world): it travels in the exchange between those two ticks — the pipeline is exactly full.

    Generator: runs this rank's compute of a tick, then yields (sends=[(buf, dst)], recvs=[(buf, src)]) for
    the exchange that follows it (empty lists when the rank has nothing to post).

    n_slots < world leaves the jobs of the slots >= n_slots out (same ticks, same links): n_slots = 1 is ONE sequence travelling
    through the stages — the single-stream latency of the pipeline, every stage idle world - 1 ticks out of world.

    stage API: first(slot, token_host | None) [rank 0; None = use the received token buffer],
    middle(slot), last(slot), last_from_first(slot) [world == 1]; buffers h_in(slot), h_out(slot), tok(slot)."""
    last = world - 1
    n_jobs = n_steps * world
    n_slots = world if n_slots is None else n_slots
    assert 1 <= n_slots <= world
    for tick in range(n_jobs + world - 1):
        j = tick - rank
        active = 0 <= j < n_jobs and j % world < n_slots
        sends, recvs = [], []
        if active:
            slot, step = j % world, j // world
            if rank == 0:
                stage.first(slot, int(first_tokens[slot]) if step == 0 else None)
            elif rank < last:
                stage.middle(slot)
            if rank == last and world > 1:
                stage.last(slot)
            elif rank == last:
                stage.last_from_first(slot)
            if rank < last:
                sends.append((stage.h_out(slot), rank + 1))
            elif step + 1 < n_steps and world > 1:
                sends.append((stage.tok(slot), 0))
        j2 = tick + 1 - rank
        if 0 <= j2 < n_jobs and world > 1 and j2 % world < n_slots:
            slot2, step2 = j2 % world, j2 // world
            if rank > 0:
                recvs.append((stage.h_in(slot2), rank - 1))
            elif step2 > 0:
                recvs.append((stage.tok(slot2), last))
        yield sends, recvs


def run_schedule(stage, comm, rank: int, world: int, n_steps: int, first_tokens, n_slots: int | None = None):
    """One rank's side of the schedule (one process per GPU).  comm API: exchange(sends, recvs); optional check() — the backend's
    asynchronous-error query (RCCL: ncclCommGetAsyncError), polled once per batch of `world` ticks; it raises."""
    check = getattr(comm, "check", None)
    for tick, (sends, recvs) in enumerate(schedule_ticks(stage, rank, world, n_steps, first_tokens, n_slots)):
        if sends or recvs:
            comm.exchange(sends, recvs)
        if check is not None and tick % world == world - 1:
            check()


def run_schedule_in_process(stages, n_steps: int, first_tokens, copy, n_slots: int | None = None):
    """All `world` stages of the pipeline driven by ONE process in lock step (stages that share a device, or a host with
    several GPUs in one address space): every tick each rank computes, then the posted sends are matched with the posted
    receives (same tick, same link) and carried out by `copy(dst_buf, src_buf)`."""
    world = len(stages)
    gens = [schedule_ticks(st, r, world, n_steps, first_tokens, n_slots) for r, st in enumerate(stages)]
    for posted in zip(*gens):
        for r, (sends, _) in enumerate(posted):
            for buf, dst in sends:
                match = [b for b, src in posted[dst][1] if src == r]
                assert len(match) == 1, f"tick: rank {r} sends to {dst}, which posted {len(match)} receives from it"
                copy(match[0], buf)
        n_s = sum(len(s_) for s_, _ in posted)
        n_r = sum(len(r_) for _, r_ in posted)
        assert n_s == n_r, "every receive of a tick has its send in the same tick"


class TorchComm:
    """torch.distributed point-to-point; one batch per tick (ncclGroupStart/End under RCCL)."""

    def __init__(self, dist, stage_through_host: bool = False, group=None):
        self.dist = dist
        self.group = group  # None: the default process group
        # gloo cannot move device tensors point-to-point: the single-GPU-box rehearsal of the
        # multi-rank path (two ranks sharing one card, backend gloo) bounces through host copies
        self.host = stage_through_host
        self._ops = {}

    def exchange(self, sends, recvs):
        d = self.dist
        if self.host:
            hs = [(t.cpu(), dst) for t, dst in sends]
            hr = [(t, t.cpu(), src) for t, src in recvs]
            ops = [d.P2POp(d.isend, c, dst) for c, dst in hs] + [d.P2POp(d.irecv, c, src) for _, c, src in hr]
            for w in d.batch_isend_irecv(ops):
                w.wait()
            for t, c, _ in hr:
                t.copy_(c)
            return
        # the (buffer, peer) pattern of a tick repeats every `world` ticks: build each P2POp list once (host time per tick
        # matters at 8 stages, where a stage's device time per tick is ~0.2 ms)
        key = (tuple((t.data_ptr(), dst) for t, dst in sends), tuple((t.data_ptr(), src) for t, src in recvs))
        ops = self._ops.get(key)
        if ops is None:
            ops = ([d.P2POp(d.isend, t, dst, group=self.group) for t, dst in sends]
                   + [d.P2POp(d.irecv, t, src, group=self.group) for t, src in recvs])
            self._ops[key] = ops
        for w in d.batch_isend_irecv(ops):
            w.wait()


class RcclComm:
    """The exchange step through the C ABI (nfai_hip_pp_*, nfai_amd/csrc/pp.hip): what a C# NFAI host would call.  The
    operations of a tick are enqueued on the stage's own stream between nfai_hip_pp_begin / _end (one RCCL group), in order
    with the stage graphs: no host synchronisation per tick."""

    def __init__(self, mgr, rank: int, world: int, unique_id: bytes):
        import ctypes as C
        from . import _lib
        self._C, self._lib, self.rank, self.world = C, _lib, rank, world
        uid = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        h = _lib.H()
        _lib.call("nfai_hip_pp_init", mgr.handle, rank, world, uid, C.byref(h))
        self.handle = h
        self._ops = {}

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _lib
        uid = (C.c_uint8 * 128)()
        _lib.call("nfai_hip_pp_unique_id", uid)
        return bytes(uid)

    def exchange(self, sends, recvs):
        """One tick: ONE native call (nfai_hip_pp_exchange: group start, sends, receives, group end on the stage stream).  The
        (buffer, peer) pattern of a tick repeats every `world` ticks, so each operation array is built once."""
        key = (tuple((t.data_ptr(), dst) for t, dst in sends), tuple((t.data_ptr(), src) for t, src in recvs))
        ops = self._ops.get(key)
        if ops is None:
            PpOp = self._lib.PpOp
            lst = [PpOp(t.data_ptr(), t.numel(), dst, 2 if t.numel() == 1 else 0, 0) for t, dst in sends]
            lst += [PpOp(t.data_ptr(), t.numel(), src, 3 if t.numel() == 1 else 1, 0) for t, src in recvs]
            ops = ((PpOp * len(lst))(*lst), len(lst))
            self._ops[key] = ops
        self._lib.call("nfai_hip_pp_exchange", self.handle, self._C.cast(ops[0], self._C.c_void_p), ops[1])

    def exchange_per_op(self, sends, recvs):
        """The same tick through the per-operation entry points (begin / send / receive / end), as a host without the array
        form would issue it."""
        C, call = self._C, self._lib.call
        call("nfai_hip_pp_begin", self.handle)
        for t, dst in sends:
            if t.numel() == 1:
                call("nfai_hip_pp_send_token", self.handle, C.c_void_p(t.data_ptr()), dst)
            else:
                call("nfai_hip_pp_send_hidden", self.handle, C.c_void_p(t.data_ptr()), t.numel(), dst)
        for t, src in recvs:
            if t.numel() == 1:
                call("nfai_hip_pp_recv_token", self.handle, C.c_void_p(t.data_ptr()), src)
            else:
                call("nfai_hip_pp_recv_hidden", self.handle, C.c_void_p(t.data_ptr()), t.numel(), src)
        call("nfai_hip_pp_end", self.handle)

    def check(self) -> None:
        """Raises NfaiHipError (naming the rank) when RCCL holds an asynchronous error for the communicator."""
        self._lib.call("nfai_hip_pp_check", self.handle)

    def wait(self, timeout_s: float) -> None:
        """Bounded synchronisation of the stage stream (nfai_hip_pp_wait): raises on an asynchronous RCCL error or at the deadline."""
        self._lib.call("nfai_hip_pp_wait", self.handle, int(min(timeout_s * 1e3, 0xFFFFFFFF)))

    def abort(self) -> None:
        self._lib.call("nfai_hip_pp_abort", self.handle)

    def info(self) -> dict:
        """RCCL's own view: ncclCommCount, ncclCommUserRank, ncclCommCuDevice + the device's PCI bus id."""
        C = self._C
        n, r, d = C.c_uint32(), C.c_uint32(), C.c_int32()
        bus = C.create_string_buffer(32)
        self._lib.call("nfai_hip_pp_info", self.handle, C.byref(n), C.byref(r), C.byref(d), bus)
        return {"nranks": n.value, "rank": r.value, "device": d.value, "pci_bus_id": bus.value.decode()}

    def close(self):
        if self.handle is not None:
            self._lib.call("nfai_hip_pp_destroy", self.handle)
            self.handle = None


class HipStage:
    """One pipeline stage on one GPU: `world` LlamaModel instances (one per in-flight sequence, each
    with its own KV cache and position) sharing one set of weights resident in HBM."""

    def __init__(self, torch, mgr, dims, layer_range, weights, n_slots, capacity, rank, world, kv_f16=False, graph=True):
        from . import _lib
        from .llama_model import LlamaModel
        self.torch, self.rank, self.world = torch, rank, world
        E = dims.E
        tens = {k: (t.data_ptr(), ty, rows, cols) for k, (t, ty, rows, cols) in weights.items()}
        d = dict(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, eps=1e-5, rope_dims=dims.D,
                 rope_base=500000.0)
        # slot 0 owns the tensors (K-quant matrices are repacked into the T16 layout once, there); the other slots alias them
        self.models = [LlamaModel(mgr, {"general.name": dims.name}, tens, capacity, layer_range=layer_range, dims=d,
                                  kv_f16=kv_f16, graph=graph)]
        for _ in range(1, n_slots):
            self.models.append(LlamaModel(mgr, {"general.name": dims.name}, tens, capacity, layer_range=layer_range, dims=d,
                                          kv_f16=kv_f16, graph=graph, share_from=self.models[0]))
        self._hin = [torch.zeros(E, device="cuda", dtype=torch.float32) for _ in range(n_slots)]
        self._hout = [torch.zeros(E, device="cuda", dtype=torch.float32) for _ in range(n_slots)]
        self._tok = [torch.zeros(1, device="cuda", dtype=torch.int32) for _ in range(n_slots)]
        self._lib = _lib

    def h_in(self, s):
        return self._hin[s]

    def h_out(self, s):
        return self._hout[s]

    def tok(self, s):
        return self._tok[s]

    def first(self, slot, token):
        m = self.models[slot]
        if token is None:
            m.TokenFromDevice(self._tok[slot].data_ptr())
            token = self._lib.TOKEN_ON_DEVICE
        if self.world == 1:
            m.StageStep(token, None, None)
        else:
            m.StageStep(token, None, self._hout[slot].data_ptr())

    def middle(self, slot):
        self.models[slot].StageStep(0, self._hin[slot].data_ptr(), self._hout[slot].data_ptr())

    def last(self, slot):
        m = self.models[slot]
        m.StageStep(0, self._hin[slot].data_ptr(), None)
        m.TokenToDevice(self._tok[slot].data_ptr())

    def last_from_first(self, slot):
        """world == 1: the only stage is first and last; `first` already ran the whole network and the argmax is the
        model's own token word, so the next step picks it up with TOKEN_ON_DEVICE (mirrored into tok(slot) for readers)."""
        self.models[slot].TokenToDevice(self._tok[slot].data_ptr())

    def dispose(self):
        for m in reversed(self.models):  # the donor (slot 0) goes last
            m.Dispose()


class PhaseWatchdog:
    """A deadline for EVERY phase of a pipeline run (the first exchanges that open RCCL's channels, the context fill, the timed
    regions).  A rank that cannot finish a phase — a peer that died, a transport that cannot be set up, a stage kernel that never
    returns — would leave every other rank waiting inside RCCL for ever: when a phase overruns, the rank prints its rank, its block
    range, the phase and the exchange in use and leaves with exit code 3 (the launcher then ends the other ranks).  A fresh exit,
    never a re-exec."""

    def __init__(self, rank: int, describe, exit_fn=None):
        self.rank, self.describe = rank, describe
        self._exit = exit_fn or (lambda code: os._exit(code))
        self._lock = threading.Lock()
        self._deadline = None
        self._name = ""
        self.fired = None
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def _run(self):
        while True:
            time.sleep(0.05)
            with self._lock:
                late = self._deadline is not None and time.monotonic() > self._deadline
                name, limit = self._name, self._limit if late else 0.0
                if late:
                    self._deadline = None
            if late:
                self.fired = name
                print(f"[rank {self.rank}] {self.describe()}: phase '{name}' did not complete within {limit:.0f} s; giving up (exit code 3)",
                      file=sys.stderr, flush=True)
                self._exit(3)

    def phase(self, name: str, limit_s: float):
        wd = self

        class _Phase:
            def __enter__(self_inner):
                with wd._lock:
                    wd._name, wd._limit, wd._deadline = name, limit_s, time.monotonic() + limit_s

            def __exit__(self_inner, *exc):
                with wd._lock:
                    wd._deadline = None
                return False

        return _Phase()


def pipeline_costs(dims, quant: str) -> tuple[float, float]:
    """(bytes one block streams per token, bytes of the lm_head) under the file type: what partition_layers balances.  Q4_K_M files
    keep attn_v / ffn_down in Q6_K on about half of the blocks (bench.tensor_type); the average block is used."""
    import bench as B
    bits = {1: 16.0, B.Q4_K: 4.5, B.Q6_K: 6.5625}
    tot = 0.0
    for name, shape in dims.shapes().items():
        if name.startswith("blk.") and len(shape) == 2:
            tot += shape[0] * shape[1] * bits[B.tensor_type(name, dims, quant)] / 8
    head = "output.weight" if not dims.tied else "token_embd.weight"
    return tot / dims.L, dims.V * dims.E * bits[B.tensor_type(head, dims, quant)] / 8


def run_bench_pipeline(args):
    """bench.py for N > 1: one process per GPU (torch.distributed.run), RCCL point-to-point.  The headline entry is the metric's
    model (Llama-3.2-3B, --quant) over `world` stages; BASELINE config 5 (Llama-3.1-8B Q4_K_M) follows as `configs[0]` in the same
    process group.  Each entry reports the aggregate of `world` sequences in flight AND the single-stream figure (one sequence
    travelling through the stages: <= the 1-GPU figure, the hops only add latency — SURVEY 8e)."""
    import torch
    import torch.distributed as dist
    from . import synth
    from ._lib import NfaiHipError
    from .hip import HipBufferManager
    import bench as B  # weight generator shared with the single-GPU bench

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # NFAI_PP_REHEARSAL=1: all ranks on device 0 with gloo (a one-GPU box cannot host two RCCL ranks);
    # exercises stages, graphs and the token hand-over, not RCCL.  Never used by the driver.
    rehearsal = os.environ.get("NFAI_PP_REHEARSAL") == "1"
    if rehearsal:
        local = 0
        # the ranks share one card: its CUs are divided between their launches, so the slices' workgroups of an attention launch are not
        # all resident at once and the granule hand-off (which waits for them) would give up — the ticket form never waits
        os.environ.setdefault("NFAI_ATTN_POLL", "0")
    torch.cuda.set_device(local)
    # Exchange step: the C ABI's own RCCL communicator (nfai_hip_pp_*, what a C# host calls) unless NFAI_PP_COMM=torch
    # (torch.distributed's NCCL binding) or the one-card rehearsal (gloo through host copies).  Host-side control traffic
    # (the RCCL unique id, the timing reduction) goes over gloo either way.
    use_cabi = not rehearsal and os.environ.get("NFAI_PP_COMM", "cabi") != "torch"
    if rehearsal or use_cabi:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("cpu:gloo,cuda:nccl", device_id=torch.device("cuda", local))
    stream = torch.cuda.Stream()
    init_limit = float(os.environ.get("NFAI_PP_INIT_TIMEOUT", "240"))
    phase_limit = float(os.environ.get("NFAI_PP_PHASE_TIMEOUT", "300"))
    where = {"text": "setting up"}
    wd = PhaseWatchdog(rank, lambda: where["text"])

    with torch.cuda.stream(stream):
        mgr = HipBufferManager(local, stream=stream.cuda_stream)
        with wd.phase("communicator set-up", init_limit):
            if use_cabi:
                # every rank learns whether ALL ranks got their communicator (a rank that failed must not leave the others in a
                # collective): if not, the exchange falls back to torch.distributed's own NCCL binding and the line says so
                comm, why = None, ""
                try:
                    box = [RcclComm.unique_id() if rank == 0 else None]
                except Exception as e:  # noqa: BLE001 - reported below, the run continues on the torch binding
                    box, why = [None], repr(e)
                dist.broadcast_object_list(box, src=0)
                if box[0] is not None:
                    try:
                        comm = RcclComm(mgr, rank, world, box[0])
                    except Exception as e:  # noqa: BLE001
                        why = repr(e)
                ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0:
                    if comm is not None:
                        comm.close()
                    print(f"[rank {rank}] nfai_hip_pp_* communicator not available on every rank ({why or 'another rank failed'}): "
                          "exchange through torch.distributed (nccl)", file=sys.stderr, flush=True)
                    use_cabi = False
                    comm = TorchComm(dist, group=dist.new_group(backend="nccl"))
            else:
                comm = TorchComm(dist, stage_through_host=rehearsal)
        exchange_name = ("nfai_hip_pp_exchange (RCCL send/recv, one group per tick on the stage stream)" if use_cabi
                         else ("gloo via host (one-card rehearsal)" if rehearsal else "torch.distributed nccl"))
        # RCCL's own record of the communicator, gathered over gloo: the line shows that RCCL saw `world` ranks on `world` devices
        rccl_view = None
        if use_cabi:
            views = [None] * world
            dist.all_gather_object(views, comm.info())
            rccl_view = {"nranks": views[0]["nranks"], "ranks": sorted(views, key=lambda v: v["rank"]),
                         "distinct_devices": len({v["pci_bus_id"] for v in views})}

        def drain(limit):
            """Bounded synchronisation of the stage stream: nfai_hip_pp_wait polls the stream AND ncclCommGetAsyncError."""
            if hasattr(comm, "wait"):
                comm.wait(limit)
            else:
                stream.synchronize()

        def give_up(phase, err):
            print(f"[rank {rank}] {where['text']}: phase '{phase}' failed: {err}; giving up (exit code 3)", file=sys.stderr, flush=True)
            try:
                if hasattr(comm, "abort"):
                    comm.abort()   # releases what can no longer complete, so that the process can leave
            except Exception:  # noqa: BLE001
                pass
            os._exit(3)

        def measure(model_name, quant, headline):
            dims = synth.BY_NAME[model_name]
            layer_bytes, head_bytes = pipeline_costs(dims, quant)
            ranges = partition_layers(dims.L, world, layer_bytes, head_bytes)
            lb, le = ranges[rank]
            first, last = rank == 0, rank == world - 1
            where["text"] = f"stage blocks [{lb},{le}) of {dims.name} ({quant}; exchange: {exchange_name})"
            weights = B.gen_weights_hbm(torch, dims, (lb, le), first, last, seed=1234 + rank, quant=quant)
            n_single = max(args.steps, 4)
            C = args.context + args.warmup + args.steps + n_single
            stage = HipStage(torch, mgr, dims, (lb, le), weights, world, C, rank, world, kv_f16=args.kv_f16, graph=not args.no_graph)
            toks = [(128000 + 17 * s) % dims.V for s in range(world)]
            phase = "?"
            try:
                # The first exchanges open the RCCL channels (xGMI peer mappings, proxy threads)
                phase = "first exchanges"
                with wd.phase(phase, init_limit):
                    run_schedule(stage, comm, rank, world, world + 1, toks)
                    drain(init_limit)
                # context fill + warmup (also instantiates the stage graphs); the sequences restart from their first tokens
                for mdl in stage.models:
                    mdl.Reset()
                phase = "context fill"
                with wd.phase(phase, phase_limit):
                    run_schedule(stage, comm, rank, world, args.context + args.warmup, toks)
                    drain(phase_limit)
                    dist.barrier()
                torch.cuda.synchronize()
                # ---- timed region 1: `world` sequences in flight (the pipeline exactly full), `steps` tokens each
                phase = "timed region (pipeline full)"
                with wd.phase(phase, phase_limit):
                    t0 = time.perf_counter()
                    run_schedule(stage, comm, rank, world, args.steps, toks)
                    drain(phase_limit)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    t = torch.tensor([dt], device="cpu", dtype=torch.float64)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    dist.barrier()
                dt = float(t.item())
                # ---- timed region 2: ONE sequence travelling through the stages (slot 0 continues): the single-stream latency
                phase = "timed region (single stream)"
                with wd.phase(phase, phase_limit):
                    run_schedule(stage, comm, rank, world, 2, toks, n_slots=1)   # the pattern's operation arrays, untimed
                    drain(phase_limit)
                    dist.barrier()
                    t0 = time.perf_counter()
                    run_schedule(stage, comm, rank, world, n_single - 2, toks, n_slots=1)
                    drain(phase_limit)
                    torch.cuda.synchronize()
                    dt1 = time.perf_counter() - t0
                    t = torch.tensor([dt1], device="cpu", dtype=torch.float64)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    dist.barrier()
                dt1 = float(t.item())
            except NfaiHipError as e:
                give_up(phase, e)
            n_tok = world * args.steps
            # bytes this rank streams per step (all its sequences), for the per-GPU achieved bandwidth
            pos_mid = args.context + args.warmup + args.steps // 2
            b_rank = sum(stage.models[0].BytesPerToken(pos_mid)[0] for _ in range(world))
            gb = torch.tensor([b_rank / 1e9], device="cpu", dtype=torch.float64)
            dist.all_reduce(gb, op=dist.ReduceOp.SUM)
            stage.dispose()
            del weights
            torch.cuda.empty_cache()
            if rank != 0:
                return None
            ach = float(gb.item()) * args.steps / dt / world
            qname = "fp16-GGUF" if quant == "f16" else "Q4_K_M-GGUF"
            return {
                "value": n_tok / dt, "unit": "tokens/s", "ms_per_step": 1e3 * dt / args.steps,
                "sequences_in_flight": world,
                "single_stream_tokens_per_s": (n_single - 2) / dt1, "single_stream_ms_per_token": 1e3 * dt1 / (n_single - 2),
                "single_stream_note": f"ONE sequence through the {world} stages ({world} stage graphs + {world} point-to-point hops per token): at batch 1 a "
                                      "layer pipeline cannot be faster than one GPU holding the whole model - capacity scales, latency does not (SURVEY 8e); "
                                      f"`value` is the aggregate of {world} independent sequences in flight",
                "config": {"workload": f"{dims.name} {qname} weights, fp32 activations + {'fp16' if args.kv_f16 else 'fp32'} KV, "
                                       f"{world} independent batch-1 greedy sequences in flight over a {world}-stage layer pipeline, "
                                       f"{args.steps} tokens each after a {args.context}-token context; then one sequence alone for {n_single - 2} tokens",
                           "parallelism": f"pp{world}", "layer_ranges": ranges, "kv_capacity": C,
                           "exchange": exchange_name, "rccl": rccl_view,
                           "async_error_poll": "ncclCommGetAsyncError once per batch of `world` ticks (nfai_hip_pp_check) and inside every bounded stream wait "
                                               "(nfai_hip_pp_wait)" if use_cabi else None,
                           "phase_deadlines_s": {"first exchanges": init_limit, "every later phase": phase_limit},
                           "multi_rank_rccl_note": None if not rehearsal else "one-card rehearsal over gloo: RCCL did not run",
                           "command": f"python bench.py --gpus {world} --model {model_name} --quant {quant} --steps {args.steps} --warmup {args.warmup}"},
                "roofline": {"bound": "hbm", "achieved": ach, "peak": B.HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / B.HBM_PEAK_GBPS,
                             "frac_of_measured_ceiling": ach / B.HBM_MEASURED_CEILING_GBPS, "traffic": None,
                             "kernel": "per-GPU average over the whole step (all kernels of the stage)"},
            }

        head = measure(args.model, args.quant, True)
        second = None
        if getattr(args, "pp_configs", "auto") != "none" and not (args.model == "llama-3.1-8b" and args.quant == "q4_k_m"):
            second = measure("llama-3.1-8b", "q4_k_m", False)   # BASELINE config 5
    if rank == 0:
        out = {
            "metric": "decode tokens/sec Llama-3.2-3B batch=1; achieved HBM GB/s vs roofline",
            "value": head["value"], "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "sequences_in_flight": world, "single_stream_tokens_per_s": head["single_stream_tokens_per_s"],
            "single_stream_ms_per_token": head["single_stream_ms_per_token"], "single_stream_note": head["single_stream_note"],
            "config": head["config"], "roofline": head["roofline"], "cpu_baseline": None,
        }
        if second is not None:
            second["baseline_config"] = f"BASELINE config 5: Llama-3.1-8B-Instruct Q4_K_M GGUF, layers pipeline-sharded across {world}xMI355X over RCCL/xGMI"
            out["configs"] = [second]
        print(json.dumps(out))
    if use_cabi:
        comm.close()
    dist.destroy_process_group()
