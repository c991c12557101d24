// Drop-in plugin (plugin boundary #2, AbstractModelFactory.TryCreate): register BEFORE LlamaModelFactory in Program.cs
// (Program.cs:16); Parser tries the factories in order and the first TryCreate that returns true wins (Parser.cs:36-42).
// NOT compiled in this repository's environment (no .NET SDK in the image).
using Microsoft.Extensions.AI;
using NFAI.Core;
using NFAI.Models;
using NFAI.Models.Llama3;
using System.Runtime.CompilerServices;

namespace NFAI.HIP;

/// <summary>≙ LlamaModelFactory (LlamaModelFactory.cs:7-45): owns the device context instead of the Vulkan instance / device.</summary>
public sealed class HipLlamaModelFactory : AbstractModelFactory
{
    private readonly HipBufferManager bufferManager;

    public HipLlamaModelFactory() : this(0) { }
    public HipLlamaModelFactory(int device) { bufferManager = new HipBufferManager(device); }

    public override void Dispose()
    {
        GC.SuppressFinalize(this);
        bufferManager.Dispose();
    }

    public override bool TryCreate(Dictionary<string, object> metadata, List<AbstractComputeCollection> tensors, ModelOptions modelOptions,
                                   out IInferenceProvider? model)
    {
        var modelFamily = metadata["general.architecture"].ToString() ?? string.Empty;
        if (modelFamily != "llama")
        {
            model = null;
            return false;
        }
        // The tensors are lazy views over a stream that Parser.Parse closes on return (Parser.cs:27): HipLlamaModel consumes
        // them all inside its constructor.
        model = new HipLlamaModel(bufferManager, metadata, tensors, modelOptions.KVCacheSize);
        return true;
    }
}

/// <summary>≙ LlamaModel (LlamaModel.cs:10-175) on libnfai_hip.so's model object: fused kernels, one hipGraph per token, the
/// whole network resident in HBM with fp16 weights kept fp16.  greedy = true swaps the reference's stochastic TopP for its own
/// ArgMax (SamplingUtils.cs:43-57), taken on the device with no logits read-back.</summary>
public sealed unsafe class HipLlamaModel : IInferenceProvider
{
    private readonly ulong model;
    private readonly Tokenizer tokenizer;
    private bool firstInput = true;

    public string ModelName { get; init; }
    public bool Greedy { get; set; }
    /// <summary>true (default): RunAsync hands every prompt token but the last to nfai_hip_llama_ingest — the batched MFMA prefill in
    /// chunks of PromptChunk tokens (fp16 operands on the matrix cores: the cache rows agree with the token-by-token path to the
    /// stated fp16 tolerance, INTEGRATION.md 3).  false: one nfai_hip_llama_decode_step per prompt token, bit for bit the M = 1 path
    /// the reference runs (LlamaModel.cs:103-126) — for parity runs.</summary>
    public bool PromptPrefill { get; set; } = true;
    /// <summary>Tokens per prefill chunk = nfai_llama_desc.max_batch (the MFMA workspace is allocated for this many rows).</summary>
    public const uint PromptChunk = 512;

    public HipLlamaModel(HipBufferManager bufferManager, Dictionary<string, object> metadata, List<AbstractComputeCollection> tensors,
                         uint contextSize = 1024u, LlamaFlags flags = LlamaFlags.None, bool referenceRopeTable = true)
    {
        ModelName = metadata["general.name"].ToString() ?? "unknown";
        tokenizer = new Tokenizer(metadata);                                                   // LlamaModel.cs:41
        var embed = tensors.FirstOrDefault(x => x.Name.Contains("token")) ?? throw new InvalidOperationException("token_embd.weight not found");
        var gate = tensors.FirstOrDefault(x => x.Name.Contains("blk.0.ffn_gate")) ?? throw new InvalidOperationException("blk.0.ffn_gate.weight not found");
        var ropeDimensions = (uint)metadata["llama.rope.dimension_count"];
        var desc = new LlamaDesc
        {
            E = (uint)embed.Shape[0], V = (uint)embed.Shape[1],
            L = (uint)metadata["llama.block_count"],
            H = (uint)metadata["llama.attention.head_count"],
            Hkv = (uint)metadata["llama.attention.head_count_kv"],
            D = (uint)metadata["llama.attention.key_length"],
            F = (uint)gate.Shape[1],
            C = contextSize,
            Eps = (float)(metadata.Where(x => x.Key.Contains("epsilon")).Select(x => x.Value).FirstOrDefault() ?? 0f),   // LlamaModel.cs:28
            RopeBase = 500000f,                                                                // hard-coded, TransformerBlock.cs:33
            RopeDims = ropeDimensions,
            // TransformerBlock.cs:66 uploads 32 frequencies whatever the head size (reference-exact); ropeDimensions / 2 = spec-correct
            RopeNFreqs = referenceRopeTable ? Math.Min(32u, ropeDimensions / 2) : ropeDimensions / 2,
            LayerBegin = 0, LayerEnd = (uint)metadata["llama.block_count"],
            Flags = (uint)flags, MaxBatch = Math.Min(contextSize, PromptChunk),
        };
        Native.Check(Native.nfai_hip_llama_create(bufferManager.Ctx, in desc, out model));
        try
        {
            foreach (var t in tensors)
            {
                if (t.Name == "rope_freqs.weight") continue;                                  // ignored by the reference (TransformerBlock.cs:33-38)
                var raw = TensorBytes.OnDisk(t);                                              // fp16 stays fp16
                ulong cols = t.Shape[0], rows = t.Shape.Length > 1 ? t.Shape[1] : 1;          // GGUF: ne0 contiguous
                fixed (byte* p = raw)
                    Native.Check(Native.nfai_hip_llama_set_tensor(model, t.Name, (int)TensorBytes.TypeOf(t), rows, cols, p));
            }
            Native.Check(Native.nfai_hip_llama_finalize(model));
        }
        catch
        {
            Native.nfai_hip_llama_destroy(model);
            throw;
        }
    }

    public async IAsyncEnumerable<ChatResponseUpdate> GetStreamingResponseAsync(IEnumerable<ChatMessage> messages, ChatOptions? options = null,
        [EnumeratorCancellation] CancellationToken cancellationToken = default)
    {
        var messageId = Guid.NewGuid().ToString();
        var userMessage = messages.FirstOrDefault(x => x.Role == ChatRole.User) ?? throw new ArgumentException("No user message found in the input messages.");
        var prompt = userMessage.Text;                                                         // LlamaModel.cs:79-80
        await foreach (var messagePart in RunAsync(prompt, cancellationToken))
        {
            yield return new ChatResponseUpdate
            {
                CreatedAt = DateTime.UtcNow,
                ModelId = ModelName,
                MessageId = messageId,
                Contents = [new TextContent(messagePart)],
            };
        }
    }

    /// <summary>≙ RunAsync (LlamaModel.cs:99-174): the prompt, then sample / feed back until EOS.  The reference runs every prompt
    /// token through the M = 1 path and keeps only the last token's logits (:103-130); here the tokens in front of the last one
    /// go through ONE native call that fills the KV cache on the MFMA prefill path (PromptPrefill), the last one through the
    /// sampled step.</summary>
    public async IAsyncEnumerable<string> RunAsync(string prompt, [EnumeratorCancellation] CancellationToken ct = default)
    {
        var tokenIds = tokenizer.Tokenize(prompt, addBos: firstInput);                          // :101
        firstInput = false;
        if (PromptPrefill && tokenIds.Count > 1)                                                // :103-126 (only the last output is sampled)
            Ingest(tokenIds.Take(tokenIds.Count - 1).ToArray());
        else
            for (int i = 0; i + 1 < tokenIds.Count; i++) Step(tokenIds[i], sample: false);
        var tk = Step(tokenIds[^1], sample: true);                                              // :128-130
        yield return tokenizer.Detokenize([tk]);
        while (tk != tokenizer.EosTokenId && !ct.IsCancellationRequested)                       // :134-173
        {
            tk = Step(tk, sample: true);
            if (tk != tokenizer.EosTokenId) yield return tokenizer.Detokenize([tk]);
        }
        await Task.CompletedTask;
    }

    /// <summary>Prompt tokens whose output is never sampled: K / V rows only, one native call (a plain method: the pinned pointer
    /// stays out of the iterator above).  NFAI_ERR_KV_FULL is raised before anything runs.</summary>
    private void Ingest(uint[] tokens)
    {
        fixed (uint* p = tokens)
            Native.Check(Native.nfai_hip_llama_ingest(model, p, (uint)tokens.Length));
    }

    private readonly uint[] candidateIds = new uint[TopK];
    private readonly float[] candidateProbs = new float[TopK];
    private const int TopK = 40;                                                                // SamplingUtils.cs:5 defaults
    private const float Temperature = 0.5f, TopPValue = 0.95f;

    /// <summary>One token through embed → blocks → norm → lm_head (one graph replay) and the next token: ArgMax on the device in
    /// greedy mode; otherwise the reference's default sampler (SamplingUtils.TopP) with its first half — values / temperature,
    /// softmax over V, stable descending order, Take(40), SamplingUtils.cs:7-13 — on the device: 328 bytes cross PCIe instead of
    /// the 513 KB of logits the reference reads back (LlamaModel.cs:128,165).</summary>
    private uint Step(uint token, bool sample)
    {
        uint argmax;
        if (Greedy || !sample)
        {
            Native.Check(Native.nfai_hip_llama_decode_step(model, token, null, &argmax));
            return argmax;
        }
        fixed (uint* ids = candidateIds)
        fixed (float* probs = candidateProbs)
            Native.Check(Native.nfai_hip_llama_decode_topk(model, token, Temperature, TopK, ids, probs));
        return TopPFromCandidates(candidateIds, candidateProbs, TopPValue, Random.Shared.NextSingle());
    }

    /// <summary>SamplingUtils.cs:14-31 on the candidates Take(topK) leaves: nucleus cut (the element that crosses topP is kept),
    /// renormalise, draw.  Same statements as the reference, so that a given `rand` selects the same token.</summary>
    internal static uint TopPFromCandidates(uint[] ids, float[] probs, float topP, float rand)
    {
        float cumulative = 0f;
        List<(uint idx, float prob)> topPList = [];
        for (int i = 0; i < ids.Length; i++)
        {
            cumulative += probs[i];
            topPList.Add((ids[i], probs[i]));
            if (cumulative >= topP) break;
        }
        float total = topPList.Sum(x => x.prob);
        float running = 0f;
        foreach (var (idx, prob) in topPList)
        {
            running += prob / total;
            if (rand < running) return idx;
        }
        return topPList[^1].idx;
    }

    /// <summary>New conversation: position 0 (the reference never resets currentToken, TransformerBlock.cs:183).</summary>
    public void Reset()
    {
        Native.Check(Native.nfai_hip_llama_reset(model));
        firstInput = true;
    }

    public void Dispose() => Native.Check(Native.nfai_hip_llama_destroy(model));               // the reference throws NotImplementedException (:70-74)
}
