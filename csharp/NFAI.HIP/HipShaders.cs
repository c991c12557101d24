// The operator surface of NFAI.Vulkan.Shaders on the HIP backend, class for class: same class names with a Hip prefix, the
// reference's public method names, constructor arguments in the reference's order minus the leading (Vk, Device) pair.  Each
// Compute* enqueues ONE hand-written gfx950 kernel through the C ABI instead of recording and fence-waiting a Vulkan dispatch
// (ShaderWrapper.cs:208-245).  Weight matrices keep their GGUF encoding in HBM (fp16 stays fp16, HipWeights).
using NFAI.Core;

namespace NFAI.HIP;

/// <summary>≙ TokenEmbedShader&lt;uint,float,float&gt; (TokenEmbedShader.cs:21-73, :108-119, :131-159).</summary>
public sealed class HipTokenEmbedShader
{
    private readonly HipBufferManager mgr;
    private readonly uint outputSize;
    public readonly HipShaderProperty<uint> inputData;
    public readonly HipWeights embeddingsData;
    public readonly HipShaderProperty<float> outputData;

    public HipTokenEmbedShader(HipBufferManager bufferManager, uint batchSize, ulong outputSize, AbstractComputeCollection embeddings)
    {
        mgr = bufferManager;
        this.outputSize = (uint)outputSize;
        inputData = new HipShaderProperty<uint>(bufferManager, batchSize) { Name = "inputData" };
        outputData = new HipShaderProperty<float>(bufferManager, outputSize) { Name = "outputData" };
        embeddingsData = new HipWeights(bufferManager, embeddings);
    }

    public HipWeights GetWeightProperty() => embeddingsData;
    public HipShaderProperty<float> GetOutputProperty() => outputData;
    public float[] GetOutputs() => outputData.GetValue();

    /// <summary>≙ Compute(ShaderProperty&lt;uint&gt;) / the token id upload of LlamaModel.cs:105-116.</summary>
    public void Compute(uint token)
    {
        inputData.SetValue([token]);
        Compute(inputData);
    }

    public void Compute(HipShaderProperty<uint> value)
    {
        if (!ReferenceEquals(value.Buffer, inputData.Buffer)) value.TransferTo(inputData);
        Native.Check(Native.nfai_hip_embed(mgr.Ctx, embeddingsData.Buffer!.Handle, (int)embeddingsData.Type, inputData.Buffer.Handle,
                                           outputData.Buffer.Handle, outputSize));
    }
}

/// <summary>≙ RMSNormShader&lt;float,float&gt; (RMSNormShader.cs:20-69, :111-122, :124-151).</summary>
public sealed class HipRMSNormShader
{
    private readonly HipBufferManager mgr;
    private readonly uint normDim;
    private readonly float epsilon;
    public readonly HipShaderProperty<float> inputData, outputData, gammaData;

    public HipRMSNormShader(HipBufferManager bufferManager, uint normalizationDimension, ComputeCollection<float> gamma, float epsilon)
    {
        mgr = bufferManager;
        normDim = normalizationDimension;
        this.epsilon = epsilon;
        inputData = new HipShaderProperty<float>(bufferManager, normDim) { Name = "inputData" };
        outputData = new HipShaderProperty<float>(bufferManager, normDim) { Name = "outputData" };
        gammaData = new HipShaderProperty<float>(bufferManager, normDim) { Name = "gammaData" };
        gammaData.SetValue(gamma);   // F32 in GGUF
    }

    public HipShaderProperty<float> GetInputProperty() => inputData;
    public HipShaderProperty<float> GetOutputProperty() => outputData;
    public float[] GetOutputs() => outputData.GetValue();

    public void Compute(HipShaderProperty<float>? value = null)
    {
        if (value != null && !ReferenceEquals(value.Buffer, inputData.Buffer)) value.TransferTo(inputData);
        Native.Check(Native.nfai_hip_rmsnorm(mgr.Ctx, inputData.Buffer.Handle, gammaData.Buffer.Handle, outputData.Buffer.Handle, normDim, epsilon));
    }
}

/// <summary>≙ MatrixMultiplyShader&lt;float,float,float&gt; with M = 1 (MatrixMultiplyShader.cs:31-131, :230-253, :255-289).  With
/// contextSize the output is a [contextSize][N] cache written at row currentCacheSize, which advances after every Compute
/// (:247-252); the reference has no bound check there, this backend throws at capacity.</summary>
public sealed class HipMatrixMultiplyShader
{
    private readonly HipBufferManager mgr;
    private readonly uint K, N, cachedContextSize;
    private readonly bool useCache;
    public uint currentCacheSize;
    public readonly HipShaderProperty<float> inputData, outputData;
    public readonly HipWeights weightData;

    public HipMatrixMultiplyShader(HipBufferManager bufferManager, uint inputRowCount, uint inputColCount, uint outputColCount,
                                   AbstractComputeCollection? weights = null, uint? contextSize = null)
    {
        if (inputRowCount != 1) throw new NotSupportedException("the reference constructs every instance with inputRowCount = 1 (TransformerBlock.cs:47-101)");
        mgr = bufferManager;
        K = inputColCount; N = outputColCount;
        useCache = contextSize is > 0;
        cachedContextSize = useCache ? contextSize!.Value : 1;
        inputData = new HipShaderProperty<float>(bufferManager, K) { Name = "inputData" };
        outputData = new HipShaderProperty<float>(bufferManager, (ulong)N * cachedContextSize) { Name = "outputData" };
        weightData = weights != null ? new HipWeights(bufferManager, weights) : new HipWeights();
    }

    public float[] GetOutputs() => outputData.GetValue();
    public HipWeights GetWeightProperty() => weightData;
    public HipShaderProperty<float> GetInputProperty() => inputData;
    public HipShaderProperty<float> GetOutputProperty() => outputData;
    public void ResetCache() => currentCacheSize = 0;                                   // (:153-159)
    public float[] GetCurrentCache() => outputData.GetValue();                          // (:313-316)

    public void Compute(HipShaderProperty<float>? value = null)
    {
        if (value != null && !ReferenceEquals(value.Buffer, inputData.Buffer)) value.TransferTo(inputData);
        if (useCache && currentCacheSize >= cachedContextSize)
            throw new NfaiHipException(4, $"KV cache row {currentCacheSize} >= capacity {cachedContextSize} (the reference writes out of bounds here, MatrixMultiplyShader.cs:248-252)");
        var offset = useCache ? (ulong)currentCacheSize * N : 0ul;
        Native.Check(Native.nfai_hip_gemv(mgr.Ctx, weightData.Buffer!.Handle, (int)weightData.Type, inputData.Buffer.Handle,
                                          outputData.Buffer.Handle, offset, N, K));
        if (useCache) currentCacheSize++;
    }
}

/// <summary>≙ RoPEShader&lt;float&gt; (RoPEShader.cs:25-131, :188-212, :231-272).  maxCacheSize == 1: one numHeads x headDim vector, out
/// of place; otherwise in place on cache row `position`.</summary>
public sealed class HipRoPEShader
{
    private readonly HipBufferManager mgr;
    private readonly uint ropeDimensions, numHeads, maxCacheSize, row, headDim;
    public readonly HipShaderProperty<float> inputData, outputData, baseFreq;

    public HipRoPEShader(HipBufferManager bufferManager, uint inputSize, uint outputSize, float[] baseFreq, uint ropeDimensions, uint numHeads,
                         uint maxCacheSize = 1)
    {
        mgr = bufferManager;
        this.ropeDimensions = ropeDimensions; this.numHeads = numHeads; this.maxCacheSize = maxCacheSize;
        row = inputSize / maxCacheSize;
        headDim = row / numHeads;
        inputData = new HipShaderProperty<float>(bufferManager, inputSize) { Name = "inputData" };
        outputData = new HipShaderProperty<float>(bufferManager, outputSize) { Name = "outputData" };
        this.baseFreq = new HipShaderProperty<float>(bufferManager, Math.Max(ropeDimensions / 2, 1)) { Name = "baseFreq" };
        var table = new float[ropeDimensions / 2];
        Array.Copy(baseFreq, table, Math.Min(baseFreq.Length, table.Length));   // entries beyond baseFreq.Length stay 0 (TransformerBlock.cs:66)
        this.baseFreq.SetValue(table);
    }

    public HipShaderProperty<float> GetInputProperty() => inputData;
    public HipShaderProperty<float> GetOutputProperty() => outputData;
    public float[] GetOutputs() => outputData.GetValue();

    public void Compute(uint position, HipShaderProperty<float>? value = null)
    {
        if (value != null && !ReferenceEquals(value.Buffer, inputData.Buffer)) value.TransferTo(inputData);
        var offset = maxCacheSize > 1 ? (ulong)position * row : 0ul;
        Native.Check(Native.nfai_hip_rope(mgr.Ctx, inputData.Buffer.Handle, offset, outputData.Buffer.Handle, offset, baseFreq.Buffer.Handle,
                                          ropeDimensions, numHeads, headDim, position));
    }
}

/// <summary>≙ AttentionScoreCalculationShader&lt;float&gt; (AttentionScoreCalculationShader.cs:22-114, :141-162, :164-206).</summary>
public sealed class HipAttentionScoreCalculationShader
{
    private readonly HipBufferManager mgr;
    private readonly uint H, Hkv, D;
    public readonly HipShaderProperty<float> queryVectors, keyCache, attentionScores;

    public HipAttentionScoreCalculationShader(HipBufferManager bufferManager, uint queryHeads, uint kvHeads, uint maxContextSize, uint headDimension)
    {
        mgr = bufferManager; H = queryHeads; Hkv = kvHeads; D = headDimension;
        queryVectors = new HipShaderProperty<float>(bufferManager, (ulong)H * D) { Name = "queryVectors" };
        keyCache = new HipShaderProperty<float>(bufferManager, (ulong)maxContextSize * Hkv * D) { Name = "keyCache" };
        attentionScores = new HipShaderProperty<float>(bufferManager, (ulong)H * maxContextSize) { Name = "attentionScores" };
    }

    public HipShaderProperty<float> GetKeyCacheProperty() => keyCache;
    public HipShaderProperty<float> GetQueryVectorsProperty() => queryVectors;
    public HipShaderProperty<float> GetAttentionScoresProperty() => attentionScores;
    public float[] GetAttentionScores() => attentionScores.GetValue();

    public void ComputeAttention(uint seqLen, HipShaderProperty<float>? queries = null, HipShaderProperty<float>? keys = null)
    {
        if (queries != null && !ReferenceEquals(queries.Buffer, queryVectors.Buffer)) queries.TransferTo(queryVectors);
        if (keys != null && !ReferenceEquals(keys.Buffer, keyCache.Buffer)) keys.TransferTo(keyCache);
        Native.Check(Native.nfai_hip_attn_scores(mgr.Ctx, queryVectors.Buffer.Handle, keyCache.Buffer.Handle, attentionScores.Buffer.Handle, H, Hkv, D, seqLen));
    }
}

/// <summary>≙ AttentionSoftmaxShader&lt;float&gt; (AttentionSoftmaxShader.cs:19-90, :117-132, :139-178).</summary>
public sealed class HipAttentionSoftmaxShader
{
    private readonly HipBufferManager mgr;
    private readonly uint H;
    private readonly float epsilon;
    public readonly HipShaderProperty<float> attentionScores, attentionWeights;

    public HipAttentionSoftmaxShader(HipBufferManager bufferManager, uint queryHeads, uint maxContextSize, uint headDimension, float epsilon = 1e-5f)
    {
        mgr = bufferManager; H = queryHeads; this.epsilon = epsilon;
        attentionScores = new HipShaderProperty<float>(bufferManager, (ulong)H * maxContextSize) { Name = "attentionScores" };
        attentionWeights = new HipShaderProperty<float>(bufferManager, (ulong)H * maxContextSize) { Name = "attentionWeights" };
    }

    public HipShaderProperty<float> GetInputProperty() => attentionScores;
    public HipShaderProperty<float> GetAttentionWeightsProperty() => attentionWeights;
    public float[] GetAttentionWeights() => attentionWeights.GetValue();
    public void SetSeqLen(uint seqLen) { }          // (:97-100) the length is an argument of ComputeSoftmax here
    public void SetSoftmaxScale(float scale) { }    // (:134-137) set but unused by the reference's GLSL: the scale is applied with the scores

    public void ComputeSoftmax(uint seqLen, HipShaderProperty<float>? scores = null)
    {
        if (scores != null && !ReferenceEquals(scores.Buffer, attentionScores.Buffer)) scores.TransferTo(attentionScores);
        Native.Check(Native.nfai_hip_attn_softmax(mgr.Ctx, attentionScores.Buffer.Handle, attentionWeights.Buffer.Handle, H, seqLen, epsilon));
    }
}

/// <summary>≙ AttentionWeightedValueSumShader&lt;float&gt; (AttentionWeightedValueSumShader.cs:21-101, :151-173, :175-216).</summary>
public sealed class HipAttentionWeightedValueSumShader
{
    private readonly HipBufferManager mgr;
    private readonly uint H, Hkv, D;
    public readonly HipShaderProperty<float> attentionWeights, valueCache, attentionOutput;

    public HipAttentionWeightedValueSumShader(HipBufferManager bufferManager, uint queryHeads, uint kvHeads, uint maxContextSize, uint headDimension)
    {
        mgr = bufferManager; H = queryHeads; Hkv = kvHeads; D = headDimension;
        attentionWeights = new HipShaderProperty<float>(bufferManager, (ulong)H * maxContextSize) { Name = "attentionWeights" };
        valueCache = new HipShaderProperty<float>(bufferManager, (ulong)maxContextSize * Hkv * D) { Name = "valueCache" };
        attentionOutput = new HipShaderProperty<float>(bufferManager, (ulong)H * D) { Name = "attentionOutput" };
    }

    public HipShaderProperty<float> GetAttentionWeights() => attentionWeights;
    public HipShaderProperty<float> GetValueCache() => valueCache;
    public HipShaderProperty<float> GetAttentionOutputProperty() => attentionOutput;
    public float[] GetAttentionOutput() => attentionOutput.GetValue();
    public void SetSeqLen(uint seqLen) { }          // (:128-131) an argument of ComputeWeightedSum here

    public void ComputeWeightedSum(uint seqLen, HipShaderProperty<float>? weights = null, HipShaderProperty<float>? values = null)
    {
        if (weights != null && !ReferenceEquals(weights.Buffer, attentionWeights.Buffer)) weights.TransferTo(attentionWeights);
        if (values != null && !ReferenceEquals(values.Buffer, valueCache.Buffer)) values.TransferTo(valueCache);
        Native.Check(Native.nfai_hip_attn_wsum(mgr.Ctx, attentionWeights.Buffer.Handle, valueCache.Buffer.Handle, attentionOutput.Buffer.Handle, H, Hkv, D, seqLen));
    }
}

/// <summary>≙ SiLUShader&lt;float&gt; (SiLUShader.cs:16-48, :92-104, :106-128).</summary>
public sealed class HipSiLUShader
{
    private readonly HipBufferManager mgr;
    private readonly uint n;
    public readonly HipShaderProperty<float> inputData, outputData;

    public HipSiLUShader(HipBufferManager bufferManager, uint elementsCount)
    {
        mgr = bufferManager; n = elementsCount;
        inputData = new HipShaderProperty<float>(bufferManager, n) { Name = "inputData" };
        outputData = new HipShaderProperty<float>(bufferManager, n) { Name = "outputData" };
    }

    public HipShaderProperty<float> GetInputProperty() => inputData;
    public HipShaderProperty<float> GetOutputProperty() => outputData;
    public float[] GetOutputs() => outputData.GetValue();

    public void Compute(HipShaderProperty<float>? value = null)
    {
        if (value != null && !ReferenceEquals(value.Buffer, inputData.Buffer)) value.TransferTo(inputData);
        Native.Check(Native.nfai_hip_silu(mgr.Ctx, inputData.Buffer.Handle, outputData.Buffer.Handle, n));
    }
}

/// <summary>≙ ElementWiseMultiplicationShader&lt;float&gt; (ElementWiseMultiplicationShader.cs:17-55, :99-119, :121-139).</summary>
public sealed class HipElementWiseMultiplicationShader
{
    private readonly HipBufferManager mgr;
    private readonly uint n;
    public readonly HipShaderProperty<float> inputDataA, inputDataB, outputData;

    public HipElementWiseMultiplicationShader(HipBufferManager bufferManager, uint elementsCount)
    {
        mgr = bufferManager; n = elementsCount;
        inputDataA = new HipShaderProperty<float>(bufferManager, n) { Name = "inputDataA" };
        inputDataB = new HipShaderProperty<float>(bufferManager, n) { Name = "inputDataB" };
        outputData = new HipShaderProperty<float>(bufferManager, n) { Name = "outputData" };
    }

    public HipShaderProperty<float> GetInputA() => inputDataA;
    public HipShaderProperty<float> GetInputB() => inputDataB;
    public HipShaderProperty<float> GetOutputProperty() => outputData;
    public float[] GetOutputs() => outputData.GetValue();

    public void Compute(HipShaderProperty<float>? valueA = null, HipShaderProperty<float>? valueB = null)
    {
        if (valueA != null && !ReferenceEquals(valueA.Buffer, inputDataA.Buffer)) valueA.TransferTo(inputDataA);
        if (valueB != null && !ReferenceEquals(valueB.Buffer, inputDataB.Buffer)) valueB.TransferTo(inputDataB);
        Native.Check(Native.nfai_hip_mul(mgr.Ctx, inputDataA.Buffer.Handle, inputDataB.Buffer.Handle, outputData.Buffer.Handle, n));
    }
}

/// <summary>≙ TransformerBlock (TransformerBlock.cs:6-214): the same 16-op chain wired with BindShaderProprty exactly as the
/// reference constructor does (:41-124) and the same Compute sequence (:127-184) — except that the two residual adds stay on the
/// device (nfai_hip_add) instead of read-back / C# add / upload (:151-161, :174-181).  This is the 1:1 operator-level surface;
/// the fast path is HipLlamaModel (fused kernels, one hipGraph per token, behind the same IInferenceProvider).</summary>
public sealed class HipTransformerBlock
{
    private readonly HipBufferManager mgr;
    private readonly HipRMSNormShader attnNormLayer, ffnNormLayer;
    private readonly HipMatrixMultiplyShader attnQueryLayer, attnKeysLayer, attnValuesLayer, attentionWeightsLayer, ffnDownLayer, ffnGateLayer, ffnUpLayer;
    private readonly HipRoPEShader ropeQueryLayer, ropeKeysLayer;
    private readonly HipAttentionScoreCalculationShader attentionScoreCalcLayer;
    private readonly HipAttentionSoftmaxShader attentionSoftmaxLayer;
    private readonly HipAttentionWeightedValueSumShader attentionWeightedValueSumLayer;
    private readonly HipSiLUShader siluLayer;
    private readonly HipElementWiseMultiplicationShader ffnProjectionLayer;
    private readonly HipShaderProperty<float> blockInput, attnResidual, blockOutput;
    private readonly uint E;
    private uint currentToken;

    public HipTransformerBlock(HipBufferManager bufferManager, List<AbstractComputeCollection> tensors, uint headDim, uint queryHeadCount, uint kvHeadCount,
                               uint contextSize, float epsilon, int index, float ropeFrequency, uint ropeDimensions, uint blockIndex,
                               int ropeTableEntries = 32)
    {
        mgr = bufferManager;
        const float ropeFreq = 500000.0f;   // hard-coded in the reference; ropeFrequency is ignored (TransformerBlock.cs:33)
        var ropeFreqs = new float[ropeDimensions / 2];
        for (var i = 0; i < ropeFreqs.Length; i++) ropeFreqs[i] = 1.0f / MathF.Pow(ropeFreq, i / (ropeDimensions / 2f));
        ropeFreqs = ropeFreqs[..Math.Min(ropeFreqs.Length, ropeTableEntries)];   // ComputeCollection<float>(memoryStream, 32, 0) (:66)

        ComputeCollection<float> T(string part) => tensors.FirstOrDefault(x => x.Name.Contains($"blk.{index}.{part}")) as ComputeCollection<float>
            ?? throw new InvalidOperationException($"Tensor with name containing 'blk.{index}.{part}' not found.");

        var attnNormCC = T("attn_norm");
        E = (uint)attnNormCC.Shape[0];
        attnNormLayer = new HipRMSNormShader(mgr, E, attnNormCC, epsilon);
        var q = T("attn_q"); var k = T("attn_k"); var v = T("attn_v");
        attnQueryLayer = new HipMatrixMultiplyShader(mgr, 1, (uint)q.Shape[0], (uint)q.Shape[1], q);
        attnQueryLayer.GetInputProperty().BindShaderProprty(attnNormLayer.GetOutputProperty());
        attnKeysLayer = new HipMatrixMultiplyShader(mgr, 1, (uint)k.Shape[0], (uint)k.Shape[1], k, contextSize);
        attnKeysLayer.GetInputProperty().BindShaderProprty(attnNormLayer.GetOutputProperty());
        attnValuesLayer = new HipMatrixMultiplyShader(mgr, 1, (uint)v.Shape[0], (uint)v.Shape[1], v, contextSize);
        attnValuesLayer.GetInputProperty().BindShaderProprty(attnNormLayer.GetOutputProperty());
        ropeQueryLayer = new HipRoPEShader(mgr, (uint)q.Shape[1], (uint)q.Shape[1], ropeFreqs, ropeDimensions, queryHeadCount);
        ropeQueryLayer.GetInputProperty().BindShaderProprty(attnQueryLayer.GetOutputProperty());
        ropeKeysLayer = new HipRoPEShader(mgr, (uint)k.Shape[1] * contextSize, (uint)k.Shape[1] * contextSize, ropeFreqs, ropeDimensions, kvHeadCount, contextSize);
        attnKeysLayer.GetOutputProperty().BindShaderProprty(ropeKeysLayer.GetOutputProperty());
        ropeKeysLayer.GetInputProperty().BindShaderProprty(attnKeysLayer.GetOutputProperty());
        var wo = T("attn_output.weight");
        attentionWeightsLayer = new HipMatrixMultiplyShader(mgr, 1, (uint)wo.Shape[0], (uint)wo.Shape[1], wo);
        var ffnNorm = T("ffn_norm");
        ffnNormLayer = new HipRMSNormShader(mgr, (uint)ffnNorm.Shape[0], ffnNorm, epsilon);
        var down = T("ffn_down"); var gate = T("ffn_gate"); var up = T("ffn_up");
        ffnDownLayer = new HipMatrixMultiplyShader(mgr, 1, (uint)down.Shape[0], (uint)down.Shape[1], down);
        ffnGateLayer = new HipMatrixMultiplyShader(mgr, 1, (uint)gate.Shape[0], (uint)gate.Shape[1], gate);
        ffnGateLayer.GetInputProperty().BindShaderProprty(ffnNormLayer.GetOutputProperty());
        ffnProjectionLayer = new HipElementWiseMultiplicationShader(mgr, (uint)down.Shape[0]);
        ffnDownLayer.GetInputProperty().BindShaderProprty(ffnProjectionLayer.GetOutputProperty());
        ffnUpLayer = new HipMatrixMultiplyShader(mgr, 1, (uint)up.Shape[0], (uint)up.Shape[1], up);
        ffnProjectionLayer.GetInputA().BindShaderProprty(ffnUpLayer.GetOutputProperty());
        ffnUpLayer.GetInputProperty().BindShaderProprty(ffnNormLayer.GetOutputProperty());
        siluLayer = new HipSiLUShader(mgr, (uint)gate.Shape[1]);
        ffnProjectionLayer.GetInputB().BindShaderProprty(siluLayer.GetOutputProperty());
        siluLayer.GetInputProperty().BindShaderProprty(ffnGateLayer.GetOutputProperty());
        attentionScoreCalcLayer = new HipAttentionScoreCalculationShader(mgr, queryHeadCount, kvHeadCount, contextSize, headDim);
        attentionScoreCalcLayer.GetQueryVectorsProperty().BindShaderProprty(ropeQueryLayer.GetOutputProperty());
        attentionScoreCalcLayer.GetKeyCacheProperty().BindShaderProprty(ropeKeysLayer.GetOutputProperty());
        attentionSoftmaxLayer = new HipAttentionSoftmaxShader(mgr, queryHeadCount, contextSize, headDim, epsilon);
        attentionSoftmaxLayer.GetInputProperty().BindShaderProprty(attentionScoreCalcLayer.GetAttentionScoresProperty());
        attentionWeightedValueSumLayer = new HipAttentionWeightedValueSumShader(mgr, queryHeadCount, kvHeadCount, contextSize, headDim);
        attentionWeightedValueSumLayer.GetValueCache().BindShaderProprty(attnValuesLayer.GetOutputProperty());
        attentionWeightedValueSumLayer.GetAttentionWeights().BindShaderProprty(attentionSoftmaxLayer.GetAttentionWeightsProperty());
        attentionWeightsLayer.GetInputProperty().BindShaderProprty(attentionWeightedValueSumLayer.GetAttentionOutputProperty());
        // the block's input is the attention norm's input (GetInputProperty, :205-208); the two residual sums get buffers of their own
        blockInput = attnNormLayer.GetInputProperty();
        attnResidual = new HipShaderProperty<float>(mgr, E) { Name = "attnResidual" };
        blockOutput = new HipShaderProperty<float>(mgr, E) { Name = "blockOutput" };
        ffnNormLayer.GetInputProperty().BindShaderProprty(attnResidual);
    }

    public HipShaderProperty<float> GetInputProperty() => blockInput;
    public HipShaderProperty<float> GetOutputProperty() => blockOutput;

    /// <summary>≙ Compute (TransformerBlock.cs:127-184), op for op.</summary>
    public void Compute(HipShaderProperty<float>? embed = null)
    {
        var pos = currentToken;
        attnNormLayer.Compute(embed);                                   // :129
        attnQueryLayer.Compute();                                       // :131
        attnKeysLayer.Compute();                                        // :133  (writes K-cache row pos)
        attnValuesLayer.Compute();                                      // :135  (writes V-cache row pos)
        ropeQueryLayer.Compute(pos);                                    // :138
        ropeKeysLayer.Compute(pos);                                     // :141  (in place on row pos)
        attentionScoreCalcLayer.ComputeAttention(pos + 1);              // :144
        attentionSoftmaxLayer.ComputeSoftmax(pos + 1);                  // :146
        attentionWeightedValueSumLayer.ComputeWeightedSum(pos + 1);     // :148
        attentionWeightsLayer.Compute();                                // :150
        Native.Check(Native.nfai_hip_add(mgr.Ctx, blockInput.Buffer.Handle, attentionWeightsLayer.GetOutputProperty().Buffer.Handle,
                                         attnResidual.Buffer.Handle, E));                                        // :151-161 on the device
        ffnNormLayer.Compute();                                         // :163
        ffnUpLayer.Compute();                                           // :165
        ffnGateLayer.Compute();                                         // :167
        siluLayer.Compute();                                            // :169
        ffnProjectionLayer.Compute();                                   // :171
        ffnDownLayer.Compute();                                         // :173
        Native.Check(Native.nfai_hip_add(mgr.Ctx, attnResidual.Buffer.Handle, ffnDownLayer.GetOutputProperty().Buffer.Handle,
                                         blockOutput.Buffer.Handle, E));                                         // :174-181 on the device
        currentToken++;                                                 // :183
    }
}
