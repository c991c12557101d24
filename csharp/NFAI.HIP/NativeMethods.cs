// P/Invoke surface of libnfai_hip.so (include/nfai_hip.h).  NOT compiled in this repository's
// environment (no .NET SDK in the image): this is the binding a maintainer adds to NFAI.
using System.Runtime.InteropServices;

namespace NFAI.HIP;

[StructLayout(LayoutKind.Sequential)]
public struct LlamaDesc
{
    public uint E, L, H, Hkv, D, F, V, C;
    public float Eps, RopeBase;
    public uint RopeDims, RopeNFreqs, LayerBegin, LayerEnd, Flags, MaxBatch;
}

public enum GgmlType { F32 = 0, F16 = 1, Q4_K = 12, Q6_K = 14 }

internal static partial class Native
{
    private const string Lib = "nfai_hip";   // libnfai_hip.so on the loader path

    [LibraryImport(Lib)] internal static partial nint nfai_hip_last_error();
    [LibraryImport(Lib)] internal static partial int nfai_hip_ctx_create(int device, out ulong ctx);
    [LibraryImport(Lib)] internal static partial int nfai_hip_ctx_destroy(ulong ctx);
    [LibraryImport(Lib)] internal static partial int nfai_hip_ctx_synchronize(ulong ctx);
    [LibraryImport(Lib)] internal static partial int nfai_hip_buf_alloc(ulong ctx, ulong bytes, out ulong buf);
    [LibraryImport(Lib)] internal static partial int nfai_hip_buf_free(ulong ctx, ulong buf);
    [LibraryImport(Lib)] internal static unsafe partial int nfai_hip_buf_upload(ulong ctx, ulong buf, ulong off, void* host, ulong bytes);
    [LibraryImport(Lib)] internal static unsafe partial int nfai_hip_buf_download(ulong ctx, ulong buf, ulong off, void* host, ulong bytes);
    [LibraryImport(Lib)] internal static partial int nfai_hip_buf_copy(ulong ctx, ulong dst, ulong doff, ulong src, ulong soff, ulong bytes);
    [LibraryImport(Lib)] internal static partial int nfai_hip_gemv(ulong ctx, ulong w, int wType, ulong x, ulong y, ulong yOff, uint n, uint k);
    [LibraryImport(Lib)] internal static partial int nfai_hip_rmsnorm(ulong ctx, ulong x, ulong g, ulong y, uint e, float eps);
    // ... one line per remaining entry point of nfai_hip.h (embed, rope, attn_*, silu, mul, add, argmax, fused ops)
    [LibraryImport(Lib)] internal static partial int nfai_hip_llama_create(ulong ctx, in LlamaDesc desc, out ulong model);
    [LibraryImport(Lib)] internal static partial int nfai_hip_llama_destroy(ulong model);
    [LibraryImport(Lib, StringMarshalling = StringMarshalling.Utf8)]
    internal static unsafe partial int nfai_hip_llama_set_tensor(ulong model, string name, int type, ulong rows, ulong cols, void* host);
    [LibraryImport(Lib)] internal static partial int nfai_hip_llama_finalize(ulong model);
    [LibraryImport(Lib)] internal static unsafe partial int nfai_hip_llama_decode_step(ulong model, uint token, float* logits, out uint argmax);
    [LibraryImport(Lib)] internal static partial int nfai_hip_llama_reset(ulong model);

    internal static void Check(int status)
    {
        if (status != 0)   // the reference throws on any non-Success VkResult (VulkanBufferManager.cs:61-87)
            throw new InvalidOperationException($"nfai_hip status {status}: {Marshal.PtrToStringUTF8(nfai_hip_last_error())}");
    }
}
