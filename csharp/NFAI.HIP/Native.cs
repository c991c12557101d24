// Hand-written part of the P/Invoke layer: the two structs that cross the C ABI by value or by pointer, the status check.
// The entry points themselves are generated from include/nfai_hip.h (NativeMethods.g.cs).
// NOT compiled in this repository's environment (no .NET SDK in the image): this is the binding a maintainer adds to NFAI.
using System.Runtime.InteropServices;

namespace NFAI.HIP;

/// <summary>nfai_llama_desc (include/nfai_hip.h): the fields LlamaModel reads from the GGUF metadata (LlamaModel.cs:23-39).</summary>
[StructLayout(LayoutKind.Sequential)]
public struct LlamaDesc
{
    public uint E, L, H, Hkv, D, F, V, C;
    public float Eps, RopeBase;
    public uint RopeDims, RopeNFreqs, LayerBegin, LayerEnd, Flags, MaxBatch;
}

/// <summary>nfai_device_info (include/nfai_hip.h).</summary>
[StructLayout(LayoutKind.Sequential)]
public unsafe struct DeviceInfo
{
    public fixed byte Name[128];
    public fixed byte Arch[64];
    public ulong TotalMemBytes;
    public uint ComputeUnits, WavefrontSize, LdsBytesPerCu, ClockKhz;
}

/// <summary>nfai_pp_op (include/nfai_hip.h): one send / receive of a pipeline tick (nfai_hip_pp_exchange).</summary>
[StructLayout(LayoutKind.Sequential)]
public unsafe struct PpOp
{
    public void* Buf;
    public uint Count, Peer, Kind, Reserved;   // Kind: 0 send hidden, 1 receive hidden, 2 send token, 3 receive token
}

/// <summary>ggml tensor type ids as stored in GGUF (Parser.cs:262-293 names the same ids).</summary>
public enum GgmlType { F32 = 0, F16 = 1, Q4_K = 12, Q6_K = 14 }

[Flags]
public enum LlamaFlags : uint { None = 0, Unfused = 1, NoGraph = 2, KvF16 = 4, Prefetch = 8, Engine = 16 }

public sealed class NfaiHipException(int status, string message) : InvalidOperationException($"nfai_hip status {status}: {message}")
{
    public int Status { get; } = status;
}

internal static unsafe partial class Native
{
    /// <summary>Every entry point returns a status; the reference throws on any non-Success VkResult
    /// (VulkanBufferManager.cs:61-87), so does this backend.</summary>
    internal static void Check(int status)
    {
        if (status != 0)
            throw new NfaiHipException(status, Marshal.PtrToStringUTF8(nfai_hip_last_error()) ?? "");
    }
}
