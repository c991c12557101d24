// HipShaderProperty<T>: ShaderProperty<T> (NFAI.Vulkan.Shaders/ShaderProperty.cs:8-263) on the HIP backend — a typed device
// buffer handle with the reference's member names.
using NFAI.Core;

namespace NFAI.HIP;

public sealed unsafe class HipShaderProperty<T> where T : unmanaged
{
    private readonly HipBufferManager manager;
    private bool owns = true;

    public HipBuffer Buffer { get; private set; }
    public string Name { get; init; } = "";
    public ulong Count { get; set; }
    public int TypeSize => sizeof(T);

    /// <summary>≙ ShaderProperty(vulkanBufferManager, transferType, count, hostVisible) (:32-43): device-local, zero-filled.</summary>
    public HipShaderProperty(HipBufferManager bufferManager, ulong count = 1)
    {
        manager = bufferManager;
        Count = count;
        Buffer = bufferManager.CreateBuffer<T>(count);
    }

    /// <summary>≙ BindShaderProprty (:95-108): free this property's own buffer and alias the other's — no reference count, as the
    /// reference.  This is how the ops of a block are chained without copies (TransformerBlock.cs:41-124).</summary>
    public void BindShaderProprty(HipShaderProperty<T> shaderProperty)
    {
        if (owns) manager.DestoryBuffer(Buffer);
        Buffer = shaderProperty.Buffer;
        Count = shaderProperty.Count;
        owns = false;
    }

    /// <summary>≙ SetValue(T[] value, int start, int count) (:110-128).</summary>
    public void SetValue(T[] value, int start, int count) => manager.UploadDeviceConstants(Buffer, value, start, count);

    /// <summary>≙ SetValue(T[] value) (:130-141).</summary>
    public void SetValue(T[] value) => manager.UploadDeviceConstants(Buffer, value, 0, value.Length);

    /// <summary>≙ SetValue(ComputeCollection&lt;T&gt; value) (:143-162): the collection as the reference uploads it (fp32 view).</summary>
    public void SetValue<TC>(ComputeCollection<TC> value) where TC : struct => manager.UploadDataToDeviceLocal(Buffer, value);

    /// <summary>≙ GetValue() (:164-182).</summary>
    public T[] GetValue() => manager.ReadDeviceBufferData<T>(Buffer, Count);

    /// <summary>≙ TransferTo(target, start) (:20-30).  The reference reads to the host and writes back; a device copy moves the
    /// same bytes.</summary>
    public void TransferTo(HipShaderProperty<T> target, int? start = null)
    {
        var n = Math.Min(Count, target.Count);
        manager.CopyBuffer(Buffer, target.Buffer, n * (ulong)sizeof(T), 0, (ulong)(start ?? 0) * (ulong)sizeof(T));
    }
}

/// <summary>A weight matrix in HBM in its GGUF encoding (≙ the weight ShaderProperty of an op after
/// SetValue(ComputeCollection), without the fp16 → fp32 widening).</summary>
public sealed class HipWeights
{
    public HipBuffer? Buffer { get; private set; }
    public GgmlType Type { get; private set; } = GgmlType.F32;
    public ulong Rows { get; private set; }
    public ulong Cols { get; private set; }

    public HipWeights() { }

    public HipWeights(HipBufferManager manager, AbstractComputeCollection tensor)
    {
        // GGUF shape = [ne0 = K contiguous, ne1 = N]: the matrix is [N][K] row-major (MatrixMultiplyShader.cs:262-288 reads W[j*K + k])
        Cols = tensor.Shape[0];
        Rows = tensor.Shape.Length > 1 ? tensor.Shape[1] : 1;
        Type = TensorBytes.TypeOf(tensor);
        Buffer = manager.UploadWeight(Type, Rows, Cols, TensorBytes.OnDisk(tensor));
    }

    /// <summary>≙ GetWeightProperty().BindShaderProprty(other) — lm_head aliases token_embd (LlamaModel.cs:64-67).</summary>
    public void BindShaderProprty(HipWeights other)
    {
        Buffer = other.Buffer; Type = other.Type; Rows = other.Rows; Cols = other.Cols;
    }
}
