"""The operator surface of NFAI.Vulkan.Shaders, class for class, on the HIP backend.

Same class and method names as the reference's ShaderWrapper subclasses, constructor arguments in
the same order minus the leading (Vk, Device) pair; each `Compute*` enqueues ONE hand-written
gfx950 kernel through the C ABI instead of recording and fence-waiting a Vulkan dispatch
(ShaderWrapper.cs:208-245).  Weights keep their GGUF encoding in HBM (fp16 stays fp16).
"""
from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import call
from .hip import HipBufferManager, ShaderProperty


def _wtype(a: np.ndarray) -> int:
    return {np.dtype(np.float16): _lib.F16, np.dtype(np.float32): _lib.F32}[a.dtype]


class _Weights:
    """Device copy of a weight matrix in its native encoding (≙ ShaderProperty.SetValue(ComputeCollection),
    ShaderProperty.cs:145-165, without the fp16->fp32 widening of AbstractComputeCollection.cs:62-77)."""

    def __init__(self, mgr: HipBufferManager, w: np.ndarray | None, ggml_type: int | None = None,
                 rows: int = 0, cols: int = 0):
        self.mgr, self.type, self.buffer, self.Count = mgr, _lib.F32, None, 0
        if w is not None:
            self.set(w, ggml_type, rows, cols)

    def set(self, w: np.ndarray, ggml_type: int | None = None, rows: int = 0, cols: int = 0):
        if ggml_type is None:
            ggml_type, (rows, cols) = _wtype(w), w.shape
        self.type = ggml_type
        self.buffer = self.mgr.UploadWeight(ggml_type, w, rows, cols)
        self.Count = rows * cols

    @property
    def handle(self):
        return self.buffer.handle

    def BindShaderProprty(self, other: "_Weights"):  # lm_head aliases token_embd (LlamaModel.cs:67)
        self.buffer, self.type, self.Count = other.buffer, other.type, other.Count


class TokenEmbedShader:
    """≙ TokenEmbedShader<uint,float,float> (TokenEmbedShader.cs:21-73, :108-119, :131-159)."""

    def __init__(self, mgr, batchSize: int, outputSize: int, embeddings: np.ndarray):
        self.mgr, self.E = mgr, int(outputSize)
        self.inputData = ShaderProperty(mgr, 1, np.uint32, "inputData")
        self.outputData = ShaderProperty(mgr, self.E, np.float32, "outputData")
        self.embeddingsData = _Weights(mgr, embeddings)

    def GetOutputProperty(self):
        return self.outputData

    def GetWeightProperty(self):
        return self.embeddingsData

    def Compute(self, token: int):
        self.inputData.SetValue(np.array([token], np.uint32))
        call("nfai_hip_embed", self.mgr.handle, self.embeddingsData.handle, self.embeddingsData.type,
             self.inputData.handle, self.outputData.handle, self.E)

    def GetOutputs(self):
        return self.outputData.GetValue()


class RMSNormShader:
    """≙ RMSNormShader<float,float> (RMSNormShader.cs:20-69, :111-122, :124-151)."""

    def __init__(self, mgr, normalizationDimension: int, gamma: np.ndarray, epsilon: float):
        self.mgr, self.E, self.eps = mgr, int(normalizationDimension), float(epsilon)
        self.inputData = ShaderProperty(mgr, self.E, name="inputData")
        self.outputData = ShaderProperty(mgr, self.E, name="outputData")
        self.gammaData = ShaderProperty(mgr, self.E, name="gammaData")
        self.gammaData.SetValue(np.asarray(gamma, np.float32))

    def GetInputProperty(self):
        return self.inputData

    def GetOutputProperty(self):
        return self.outputData

    def Compute(self, value: ShaderProperty | None = None):
        if value is not None and value.buffer is not self.inputData.buffer:
            value.TransferTo(self.inputData)
        call("nfai_hip_rmsnorm", self.mgr.handle, self.inputData.handle, self.gammaData.handle,
             self.outputData.handle, self.E, self.eps)

    def GetOutputs(self):
        return self.outputData.GetValue()


class MatrixMultiplyShader:
    """≙ MatrixMultiplyShader<float,float,float>, M = 1 (MatrixMultiplyShader.cs:31-131, :230-253,
    :255-289).  With `contextSize` the output is a [contextSize][N] cache written at row
    `currentCacheSize`, which advances after every Compute (:247-252)."""

    def __init__(self, mgr, inputRowCount: int, inputColCount: int, outputColCount: int,
                 weights: np.ndarray | None = None, contextSize: int | None = None):
        assert inputRowCount == 1, "the reference constructs every instance with M = 1 (TransformerBlock.cs:47-101)"
        self.mgr, self.K, self.N = mgr, int(inputColCount), int(outputColCount)
        self.useCache = bool(contextSize)
        self.cachedContextSize = int(contextSize) if self.useCache else 1
        self.currentCacheSize = 0
        self.inputData = ShaderProperty(mgr, self.K, name="inputData")
        self.outputData = ShaderProperty(mgr, self.N * self.cachedContextSize, name="outputData")
        self.weightData = _Weights(mgr, weights)

    def GetInputProperty(self):
        return self.inputData

    def GetOutputProperty(self):
        return self.outputData

    def GetWeightProperty(self):
        return self.weightData

    def ResetCache(self):  # (:153-159)
        self.currentCacheSize = 0

    def Compute(self, value: ShaderProperty | None = None):
        if value is not None and value.buffer is not self.inputData.buffer:
            value.TransferTo(self.inputData)
        if self.useCache and self.currentCacheSize >= self.cachedContextSize:
            # the reference keeps counting and writes out of bounds (:248-252); hard error here
            raise _lib.KVCacheFull(_lib.ERR_KV_FULL, f"cache row {self.currentCacheSize} >= capacity {self.cachedContextSize}")
        off = self.currentCacheSize * self.N if self.useCache else 0
        call("nfai_hip_gemv", self.mgr.handle, self.weightData.handle, self.weightData.type, self.inputData.handle,
             self.outputData.handle, off, self.N, self.K)
        if self.useCache:
            self.currentCacheSize += 1

    def GetOutputs(self):
        return self.outputData.GetValue()


class RoPEShader:
    """≙ RoPEShader<float> (RoPEShader.cs:25-131, :188-212, :231-272).  maxCacheSize == 1: one
    n_heads x head_dim vector, out of place; otherwise in place on cache row `position`."""

    def __init__(self, mgr, inputSize: int, outputSize: int, baseFreq: np.ndarray, ropeDimensions: int,
                 numHeads: int, maxCacheSize: int = 1):
        self.mgr, self.rope_dims, self.n_heads, self.maxCacheSize = mgr, int(ropeDimensions), int(numHeads), int(maxCacheSize)
        self.row = int(inputSize) // self.maxCacheSize
        self.head_dim = self.row // self.n_heads
        self.inputData = ShaderProperty(mgr, inputSize, name="inputData")
        self.outputData = ShaderProperty(mgr, inputSize, name="outputData")
        self.baseFreq = ShaderProperty(mgr, max(self.rope_dims // 2, 1), name="baseFreq")
        f = np.zeros(self.rope_dims // 2, np.float32)
        bf = np.asarray(baseFreq, np.float32)
        f[:bf.size] = bf[:f.size]  # the reference uploads only baseFreq.Length entries (TransformerBlock.cs:66)
        self.baseFreq.SetValue(f)

    def GetInputProperty(self):
        return self.inputData

    def GetOutputProperty(self):
        return self.outputData

    def Compute(self, position: int, value: ShaderProperty | None = None):
        if value is not None and value.buffer is not self.inputData.buffer:
            value.TransferTo(self.inputData)
        off = position * self.row if self.maxCacheSize > 1 else 0
        call("nfai_hip_rope", self.mgr.handle, self.inputData.handle, off, self.outputData.handle, off,
             self.baseFreq.handle, self.rope_dims, self.n_heads, self.head_dim, position)

    def GetOutputs(self):
        return self.outputData.GetValue()


class AttentionScoreCalculationShader:
    """≙ AttentionScoreCalculationShader<float> (…ScoreCalculationShader.cs:22-114, :141-162, :164-206)."""

    def __init__(self, mgr, queryHeads: int, kvHeads: int, maxContextSize: int, headDimension: int):
        self.mgr, self.H, self.Hkv, self.C, self.D = mgr, queryHeads, kvHeads, maxContextSize, headDimension
        self.queryVectors = ShaderProperty(mgr, self.H * self.D, name="queryVectors")
        self.keyCache = ShaderProperty(mgr, self.C * self.Hkv * self.D, name="keyCache")
        self.attentionScores = ShaderProperty(mgr, self.H * self.C, name="attentionScores")

    def GetQueryVectorsProperty(self):
        return self.queryVectors

    def GetKeyCacheProperty(self):
        return self.keyCache

    def GetAttentionScoresProperty(self):
        return self.attentionScores

    def ComputeAttention(self, seqLen: int):
        call("nfai_hip_attn_scores", self.mgr.handle, self.queryVectors.handle, self.keyCache.handle,
             self.attentionScores.handle, self.H, self.Hkv, self.D, seqLen)


class AttentionSoftmaxShader:
    """≙ AttentionSoftmaxShader<float> (AttentionSoftmaxShader.cs:19-90, :117-132, :139-178)."""

    def __init__(self, mgr, queryHeads: int, maxContextSize: int, headDimension: int, epsilon: float = 1e-5):
        self.mgr, self.H, self.C, self.eps = mgr, queryHeads, maxContextSize, float(epsilon)
        self.attentionScores = ShaderProperty(mgr, self.H * self.C, name="attentionScores")
        self.attentionWeights = ShaderProperty(mgr, self.H * self.C, name="attentionWeights")

    def GetInputProperty(self):
        return self.attentionScores

    def GetAttentionWeightsProperty(self):
        return self.attentionWeights

    def ComputeSoftmax(self, seqLen: int):
        call("nfai_hip_attn_softmax", self.mgr.handle, self.attentionScores.handle, self.attentionWeights.handle,
             self.H, seqLen, self.eps)


class AttentionWeightedValueSumShader:
    """≙ AttentionWeightedValueSumShader<float> (…ValueSumShader.cs:21-101, :151-173, :175-216)."""

    def __init__(self, mgr, queryHeads: int, kvHeads: int, maxContextSize: int, headDimension: int):
        self.mgr, self.H, self.Hkv, self.C, self.D = mgr, queryHeads, kvHeads, maxContextSize, headDimension
        self.attentionWeights = ShaderProperty(mgr, self.H * self.C, name="attentionWeights")
        self.valueCache = ShaderProperty(mgr, self.C * self.Hkv * self.D, name="valueCache")
        self.attentionOutput = ShaderProperty(mgr, self.H * self.D, name="attentionOutput")

    def GetValueCache(self):
        return self.valueCache

    def GetAttentionWeights(self):
        return self.attentionWeights

    def GetAttentionOutputProperty(self):
        return self.attentionOutput

    def ComputeWeightedSum(self, seqLen: int):
        call("nfai_hip_attn_wsum", self.mgr.handle, self.attentionWeights.handle, self.valueCache.handle,
             self.attentionOutput.handle, self.H, self.Hkv, self.D, seqLen)


class SiLUShader:
    """≙ SiLUShader<float> (SiLUShader.cs:16-48, :92-104, :106-128)."""

    def __init__(self, mgr, elementsCount: int):
        self.mgr, self.n = mgr, int(elementsCount)
        self.inputData = ShaderProperty(mgr, self.n, name="inputData")
        self.outputData = ShaderProperty(mgr, self.n, name="outputData")

    def GetInputProperty(self):
        return self.inputData

    def GetOutputProperty(self):
        return self.outputData

    def Compute(self):
        call("nfai_hip_silu", self.mgr.handle, self.inputData.handle, self.outputData.handle, self.n)


class ElementWiseMultiplicationShader:
    """≙ ElementWiseMultiplicationShader<float> (ElementWiseMultiplicationShader.cs:17-55, :99-119, :121-139)."""

    def __init__(self, mgr, elementsCount: int):
        self.mgr, self.n = mgr, int(elementsCount)
        self.inputDataA = ShaderProperty(mgr, self.n, name="inputDataA")
        self.inputDataB = ShaderProperty(mgr, self.n, name="inputDataB")
        self.outputData = ShaderProperty(mgr, self.n, name="outputData")

    def GetInputA(self):
        return self.inputDataA

    def GetInputB(self):
        return self.inputDataB

    def GetOutputProperty(self):
        return self.outputData

    def Compute(self):
        call("nfai_hip_mul", self.mgr.handle, self.inputDataA.handle, self.inputDataB.handle, self.outputData.handle, self.n)


class TransformerBlock:
    """≙ TransformerBlock (TransformerBlock.cs:6-214): the same 16-op chain wired with
    BindShaderProprty exactly as the reference constructor does (:41-124), and the same Compute
    sequence (:127-184) — except that the two residual adds stay on the device (nfai_hip_add)
    instead of read-back / C# add / upload (:151-161, :174-181).

    `tensors`: dict GGUF-name -> ndarray (float16 / float32).  This is the 1:1 parity surface; the
    fast path is `nfai_amd.llama_model.LlamaModel` (fused kernels + hipGraph behind the same
    LlamaModel / IInferenceProvider shape).
    """

    def __init__(self, mgr, tensors: dict, headDim: int, queryHeadCount: int, kvHeadCount: int, contextSize: int,
                 epsilon: float, index: int, ropeFrequency: float, ropeDimensions: int, blockIndex: int,
                 ropeTableEntries: int | None = 32):
        ropeFreq = np.float32(500000.0)  # hard-coded in the reference; `ropeFrequency` is ignored (:33)
        half = ropeDimensions // 2
        i = np.arange(half, dtype=np.float32)
        ropeFreqs = (np.float32(1.0) / np.power(ropeFreq, i / np.float32(half), dtype=np.float32)).astype(np.float32)
        if ropeTableEntries is not None:
            ropeFreqs = ropeFreqs[:ropeTableEntries]  # ComputeCollection<float>(memoryStream, 32, 0) (:66)
        t = lambda s: tensors[f"blk.{index}.{s}.weight"]
        self.blockIndex, self.currentToken, self.mgr = blockIndex, 0, mgr
        E = t("attn_norm").shape[0]
        self.attnNormLayer = RMSNormShader(mgr, E, t("attn_norm"), epsilon)
        wq, wk, wv = t("attn_q"), t("attn_k"), t("attn_v")
        self.attnQueryLayer = MatrixMultiplyShader(mgr, 1, wq.shape[1], wq.shape[0], wq)
        self.attnQueryLayer.GetInputProperty().BindShaderProprty(self.attnNormLayer.GetOutputProperty())
        self.attnKeysLayer = MatrixMultiplyShader(mgr, 1, wk.shape[1], wk.shape[0], wk, contextSize)
        self.attnKeysLayer.GetInputProperty().BindShaderProprty(self.attnNormLayer.GetOutputProperty())
        self.attnValuesLayer = MatrixMultiplyShader(mgr, 1, wv.shape[1], wv.shape[0], wv, contextSize)
        self.attnValuesLayer.GetInputProperty().BindShaderProprty(self.attnNormLayer.GetOutputProperty())
        self.ropeQueryLayer = RoPEShader(mgr, wq.shape[0], wq.shape[0], ropeFreqs, ropeDimensions, queryHeadCount)
        self.ropeQueryLayer.GetInputProperty().BindShaderProprty(self.attnQueryLayer.GetOutputProperty())
        self.ropeKeysLayer = RoPEShader(mgr, wk.shape[0] * contextSize, wk.shape[0] * contextSize, ropeFreqs,
                                        ropeDimensions, kvHeadCount, contextSize)
        self.attnKeysLayer.GetOutputProperty().BindShaderProprty(self.ropeKeysLayer.GetOutputProperty())
        self.ropeKeysLayer.GetInputProperty().BindShaderProprty(self.attnKeysLayer.GetOutputProperty())
        wo = t("attn_output")
        self.attentionWeightsLayer = MatrixMultiplyShader(mgr, 1, wo.shape[1], wo.shape[0], wo)
        self.ffnNormLayer = RMSNormShader(mgr, E, t("ffn_norm"), epsilon)
        self.ffnNormLayer.GetInputProperty().BindShaderProprty(self.attentionWeightsLayer.GetOutputProperty())
        wd, wg, wu = t("ffn_down"), t("ffn_gate"), t("ffn_up")
        self.ffnDownLayer = MatrixMultiplyShader(mgr, 1, wd.shape[1], wd.shape[0], wd)
        self.ffnGateLayer = MatrixMultiplyShader(mgr, 1, wg.shape[1], wg.shape[0], wg)
        self.ffnGateLayer.GetInputProperty().BindShaderProprty(self.ffnNormLayer.GetOutputProperty())
        self.ffnProjectionLayer = ElementWiseMultiplicationShader(mgr, wd.shape[1])
        self.ffnDownLayer.GetInputProperty().BindShaderProprty(self.ffnProjectionLayer.GetOutputProperty())
        self.ffnUpLayer = MatrixMultiplyShader(mgr, 1, wu.shape[1], wu.shape[0], wu)
        self.ffnProjectionLayer.GetInputA().BindShaderProprty(self.ffnUpLayer.GetOutputProperty())
        self.ffnUpLayer.GetInputProperty().BindShaderProprty(self.ffnNormLayer.GetOutputProperty())
        self.siluLayer = SiLUShader(mgr, wg.shape[0])
        self.ffnProjectionLayer.GetInputB().BindShaderProprty(self.siluLayer.GetOutputProperty())
        self.siluLayer.GetInputProperty().BindShaderProprty(self.ffnGateLayer.GetOutputProperty())
        self.attentionScoreCalcLayer = AttentionScoreCalculationShader(mgr, queryHeadCount, kvHeadCount, contextSize, headDim)
        self.attentionScoreCalcLayer.GetQueryVectorsProperty().BindShaderProprty(self.ropeQueryLayer.GetOutputProperty())
        self.attentionScoreCalcLayer.GetKeyCacheProperty().BindShaderProprty(self.ropeKeysLayer.GetOutputProperty())
        self.attentionSoftmaxLayer = AttentionSoftmaxShader(mgr, queryHeadCount, contextSize, headDim, epsilon)
        self.attentionSoftmaxLayer.GetInputProperty().BindShaderProprty(self.attentionScoreCalcLayer.GetAttentionScoresProperty())
        self.attentionWeightedValueSumLayer = AttentionWeightedValueSumShader(mgr, queryHeadCount, kvHeadCount, contextSize, headDim)
        self.attentionWeightedValueSumLayer.GetValueCache().BindShaderProprty(self.attnValuesLayer.GetOutputProperty())
        self.attentionWeightedValueSumLayer.GetAttentionWeights().BindShaderProprty(self.attentionSoftmaxLayer.GetAttentionWeightsProperty())
        self.attentionWeightsLayer.GetInputProperty().BindShaderProprty(self.attentionWeightedValueSumLayer.GetAttentionOutputProperty())
        self.E = E

    def Compute(self, embed: ShaderProperty | None = None):
        self.attnNormLayer.Compute(embed)
        self.attnQueryLayer.Compute()
        self.attnKeysLayer.Compute()
        self.attnValuesLayer.Compute()
        self.ropeQueryLayer.Compute(self.currentToken)
        self.ropeKeysLayer.Compute(self.currentToken)
        self.attentionScoreCalcLayer.ComputeAttention(self.currentToken + 1)
        self.attentionSoftmaxLayer.ComputeSoftmax(self.currentToken + 1)
        self.attentionWeightedValueSumLayer.ComputeWeightedSum(self.currentToken + 1)
        self.attentionWeightsLayer.Compute()
        proj = self.attentionWeightsLayer.GetOutputProperty()
        # attnFinal = input + attnWeightsOutput, stored back into the Wo output property (:153-161)
        call("nfai_hip_add", self.mgr.handle, self.attnNormLayer.GetInputProperty().handle, proj.handle, proj.handle, self.E)
        self.ffnNormLayer.Compute()
        self.ffnUpLayer.Compute()
        self.ffnGateLayer.Compute()
        self.siluLayer.Compute()
        self.ffnProjectionLayer.Compute()
        self.ffnDownLayer.Compute()
        down = self.ffnDownLayer.GetOutputProperty()
        # finalResidual = attnFinal + ffnDownOutput, stored into the Wdown output property (:176-181)
        call("nfai_hip_add", self.mgr.handle, proj.handle, down.handle, down.handle, self.E)
        self.currentToken += 1

    def GetInputProperty(self):
        return self.attnNormLayer.inputData

    def GetOutputProperty(self):
        return self.ffnDownLayer.GetOutputProperty()
