// The layer pipeline across the GPUs of one node from a C# host: one process (or thread with its own HipBufferManager) per GPU
// owns a contiguous range of blocks (LlamaDesc.LayerBegin / LayerEnd) and exchanges the hidden state with its neighbours through
// libnfai_hip.so's RCCL point-to-point entry points (nfai_hip_pp_*).  The reference is single-device (it takes the last
// enumerated Vulkan device, VulkanHelper.cs:149-150): this has no counterpart there.  NOT compiled in this repository.
namespace NFAI.HIP;

public sealed unsafe class HipPipeline : IDisposable
{
    private ulong handle;
    public uint Rank { get; }
    public uint World { get; }

    /// <summary>Rank 0 creates the 128-byte id and hands it to every rank (any host channel).</summary>
    public static byte[] CreateUniqueId()
    {
        var id = new byte[128];
        fixed (byte* p = id) Native.Check(Native.nfai_hip_pp_unique_id(p));
        return id;
    }

    public HipPipeline(HipBufferManager bufferManager, uint rank, uint world, byte[] uniqueId)
    {
        if (uniqueId.Length != 128) throw new ArgumentException("the RCCL unique id is 128 bytes", nameof(uniqueId));
        Rank = rank; World = world;
        fixed (byte* p = uniqueId) Native.Check(Native.nfai_hip_pp_init(bufferManager.Ctx, rank, world, p, out handle));
    }

    /// <summary>The operations of one tick of the schedule are posted between Begin and End (one RCCL group): both ends of a
    /// link post in the same tick.</summary>
    public void Begin() => Native.Check(Native.nfai_hip_pp_begin(handle));
    public void End() => Native.Check(Native.nfai_hip_pp_end(handle));
    public void SendHidden(nint hiddenDev, uint nFloats, uint peer) => Native.Check(Native.nfai_hip_pp_send_hidden(handle, (void*)hiddenDev, nFloats, peer));
    public void RecvHidden(nint hiddenDev, uint nFloats, uint peer) => Native.Check(Native.nfai_hip_pp_recv_hidden(handle, (void*)hiddenDev, nFloats, peer));
    public void SendToken(nint tokenDev, uint peer) => Native.Check(Native.nfai_hip_pp_send_token(handle, (void*)tokenDev, peer));
    public void RecvToken(nint tokenDev, uint peer) => Native.Check(Native.nfai_hip_pp_recv_token(handle, (void*)tokenDev, peer));
    /// <summary>A whole tick in one native call (group start, the operations, group end).</summary>
    public void Exchange(ReadOnlySpan<PpOp> ops)
    {
        fixed (PpOp* p = ops) Native.Check(Native.nfai_hip_pp_exchange(handle, p, (uint)ops.Length));
    }
    /// <summary>What RCCL reports for this communicator: (ranks, this rank, device ordinal, PCI bus id).</summary>
    public (uint NRanks, uint Rank, int Device, string BusId) Info()
    {
        uint n, r; int d;
        var bus = stackalloc byte[32];
        Native.Check(Native.nfai_hip_pp_info(handle, &n, &r, &d, bus));
        return (n, r, d, System.Runtime.InteropServices.Marshal.PtrToStringUTF8((nint)bus) ?? "");
    }
    /// <summary>Throws when RCCL holds an asynchronous error for this communicator (a dead peer, a broken link): poll it once per
    /// batch of ticks.</summary>
    public void Check() => Native.Check(Native.nfai_hip_pp_check(handle));
    /// <summary>Bounded synchronisation of the stage stream: returns when it is idle, throws (naming the rank) on an asynchronous
    /// RCCL error or when the deadline passes.</summary>
    public void Wait(TimeSpan deadline) => Native.Check(Native.nfai_hip_pp_wait(handle, (uint)Math.Min(deadline.TotalMilliseconds, uint.MaxValue)));
    /// <summary>ncclCommAbort: releases operations that can no longer complete so that the process can leave.</summary>
    public void Abort() => Native.Check(Native.nfai_hip_pp_abort(handle));
    public void BroadcastToken(nint tokenDev, uint root) => Native.Check(Native.nfai_hip_pp_bcast_token(handle, (void*)tokenDev, root));

    public void Dispose()
    {
        if (handle == 0) return;
        Native.Check(Native.nfai_hip_pp_destroy(handle));
        handle = 0;
    }
}
