// HipBufferManager: the backend surface of NFAI.Vulkan/VulkanBufferManager.cs (public members :37-499) on libnfai_hip.so.
// Method names follow the reference, spellings included (DestoryBuffer).  One instance per GPU, not thread-safe — as the reference
// (one queue, one blocking submit at a time, :474-494).  There is no CPU fallback: without a gfx950 device the constructor throws.
using NFAI.Core;
using System.Runtime.InteropServices;

namespace NFAI.HIP;

/// <summary>A device allocation (≙ the Buffer + DeviceMemory pair VulkanBufferManager.CreateBuffer hands out, :42-88).</summary>
public sealed class HipBuffer(HipBufferManager manager, ulong handle, ulong bytes, bool owned = true)
{
    public HipBufferManager Manager { get; } = manager;
    public ulong Handle { get; internal set; } = handle;
    public ulong Bytes { get; } = bytes;
    internal bool Owned { get; } = owned;
}

public sealed unsafe class HipBufferManager : IDisposable
{
    internal ulong Ctx;
    public int Device { get; }
    public DeviceInfo Info { get; }

    /// <summary>≙ VulkanHelper.CreateVulkanInstance / PickPhysicalDevice / CreateLogicalDevice + the VulkanBufferManager
    /// constructor (LlamaModelFactory.cs:15-22).</summary>
    public HipBufferManager(int device = 0)
    {
        Device = device;
        Native.Check(Native.nfai_hip_ctx_create(device, out Ctx));
        Native.Check(Native.nfai_hip_ctx_device_info(Ctx, out var info));
        Info = info;
    }

    /// <summary>≙ CreateBuffer&lt;T&gt;(count, usage, props, out buf, out mem) (:42-88); zero-filled.</summary>
    public HipBuffer CreateBuffer<T>(ulong count) where T : unmanaged
    {
        var bytes = count * (ulong)sizeof(T);
        Native.Check(Native.nfai_hip_buf_alloc(Ctx, bytes, out var h));
        return new HipBuffer(this, h, bytes);
    }

    /// <summary>≙ DestoryBuffer (:90-103).</summary>
    public void DestoryBuffer(HipBuffer buffer)
    {
        if (buffer.Handle == 0) return;
        Native.Check(Native.nfai_hip_buf_free(Ctx, buffer.Handle));
        buffer.Handle = 0;
    }

    /// <summary>≙ UploadDeviceConstants&lt;T&gt;(ref buf, T[], start, count) / UploadConstants (:196-244, :246-268).</summary>
    public void UploadDeviceConstants<T>(HipBuffer buffer, T[] data, int start = 0, int? count = null) where T : unmanaged
    {
        var n = count ?? data.Length;
        fixed (T* p = data)
            Native.Check(Native.nfai_hip_buf_upload(Ctx, buffer.Handle, (ulong)start * (ulong)sizeof(T), p, (ulong)n * (ulong)sizeof(T)));
    }

    /// <summary>≙ UploadDataToDeviceLocal&lt;T&gt;(buf, ComputeCollection&lt;T&gt;) (:105-125, :130-194): the collection's bytes as the
    /// reference would upload them (fp16 tensors arrive widened to fp32, AbstractComputeCollection.cs:62-77), streamed batch by
    /// batch.  Weight matrices do NOT come this way: see HipWeights, which keeps fp16 as fp16 in HBM.</summary>
    public void UploadDataToDeviceLocal<T>(HipBuffer buffer, ComputeCollection<T> data) where T : struct
    {
        ulong off = 0;
        foreach (var batch in data.GetDataRaw())
        {
            fixed (byte* p = batch)
                Native.Check(Native.nfai_hip_buf_upload(Ctx, buffer.Handle, off, p, (ulong)batch.Length));
            off += (ulong)batch.Length;
        }
    }

    /// <summary>≙ ReadDeviceBufferData&lt;T&gt;(ref buf, count) / ReadBufferData (:283-303, :270-281).</summary>
    public T[] ReadDeviceBufferData<T>(HipBuffer buffer, ulong count, ulong start = 0) where T : unmanaged
    {
        var result = new T[count];
        fixed (T* p = result)
            Native.Check(Native.nfai_hip_buf_download(Ctx, buffer.Handle, start * (ulong)sizeof(T), p, count * (ulong)sizeof(T)));
        return result;
    }

    /// <summary>≙ CopyBuffer(src, dst, size) (:305-318).</summary>
    public void CopyBuffer(HipBuffer src, HipBuffer dst, ulong sizeBytes, ulong srcOffset = 0, ulong dstOffset = 0)
        => Native.Check(Native.nfai_hip_buf_copy(Ctx, dst.Handle, dstOffset, src.Handle, srcOffset, sizeBytes));

    /// <summary>≙ vkQueueWaitIdle (:334).  Operator calls only enqueue; this waits for the device.</summary>
    public void Synchronize() => Native.Check(Native.nfai_hip_ctx_synchronize(Ctx));

    /// <summary>A weight tensor in its GGUF encoding (fp16 stays fp16; Q4_K / Q6_K blocks stay quantised) → HBM.</summary>
    public HipBuffer UploadWeight(GgmlType type, ulong rows, ulong cols, byte[] raw)
    {
        fixed (byte* p = raw)
        {
            Native.Check(Native.nfai_hip_weight_upload(Ctx, (int)type, rows, cols, p, out var h));
            return new HipBuffer(this, h, (ulong)raw.LongLength);
        }
    }

    /// <summary>≙ Dispose (:499-509).</summary>
    public void Dispose()
    {
        if (Ctx == 0) return;
        Native.Check(Native.nfai_hip_ctx_destroy(Ctx));
        Ctx = 0;
        GC.SuppressFinalize(this);
    }
}

/// <summary>The bytes of a tensor as they sit in the GGUF file.  The reference's public accessor,
/// ComputeCollection&lt;T&gt;.GetDataRaw() (ComputeCollection.cs:41-44), widens fp16 to fp32 on the fly
/// (AbstractComputeCollection.cs:62-77); every widened value is exactly representable in fp16, so narrowing it back is
/// lossless and recovers the on-disk bytes without touching the (private) stream.</summary>
public static class TensorBytes
{
    public static GgmlType TypeOf(AbstractComputeCollection t) => t.TypeSize == 2 ? GgmlType.F16 : GgmlType.F32;

    public static byte[] OnDisk(AbstractComputeCollection t)
    {
        var cc = t as ComputeCollection<float> ?? throw new NotSupportedException(
            $"tensor {t.Name}: the reference's parser yields ComputeCollection<float> for F32 and F16 tensors (Parser.cs:61-110) and throws for every other type");
        var total = checked((long)(cc.Length * cc.TypeSize));
        var result = new byte[total];
        long off = 0;
        foreach (var batch in cc.GetDataRaw())
        {
            if (cc.TypeSize == 2)
            {
                var floats = MemoryMarshal.Cast<byte, float>(batch);
                var halves = MemoryMarshal.Cast<byte, Half>(result.AsSpan((int)off, floats.Length * 2));
                for (var i = 0; i < floats.Length; i++) halves[i] = (Half)floats[i];   // exact: the float came from a Half
                off += floats.Length * 2;
            }
            else
            {
                batch.CopyTo(result.AsSpan((int)off));
                off += batch.Length;
            }
        }
        if (off != total) throw new InvalidDataException($"tensor {t.Name}: read {off} of {total} bytes");
        return result;
    }
}
