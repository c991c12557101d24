// Drop-in plugin: register BEFORE LlamaModelFactory in Program.cs (Program.cs:16); Parser tries the
// factories in order and the first TryCreate that returns true wins (Parser.cs:36-42).
using Microsoft.Extensions.AI;
using NFAI.Core;
using NFAI.Models;
using NFAI.Models.Llama3;
using System.Runtime.CompilerServices;

namespace NFAI.HIP;

public sealed class HipBufferManager : IDisposable          // ≙ NFAI.Vulkan.VulkanBufferManager
{
    internal readonly ulong Ctx;
    public HipBufferManager(int device = 0) { Native.Check(Native.nfai_hip_ctx_create(device, out Ctx)); }
    public void Dispose() => Native.Check(Native.nfai_hip_ctx_destroy(Ctx));
}

public sealed class HipLlamaModelFactory : AbstractModelFactory   // ≙ LlamaModelFactory.cs:7-45
{
    private readonly HipBufferManager mgr = new();
    public override void Dispose() => mgr.Dispose();

    public override bool TryCreate(Dictionary<string, object> metadata, List<AbstractComputeCollection> tensors,
                                   ModelOptions modelOptions, out IInferenceProvider? model)
    {
        model = null;
        if ((metadata["general.architecture"].ToString() ?? "") != "llama") return false;
        // The tensors are lazy views over a stream that Parser.Parse closes on return (Parser.cs:27):
        // consume them here.  (Raw fp16 bytes are uploaded as-is; AbstractComputeCollection.GetDataRaw
        // widens fp16 -> fp32, so HipLlamaModel reads the stream range [offset, offset + Length*2) itself.)
        model = new HipLlamaModel(mgr, metadata, tensors, modelOptions.KVCacheSize);
        return true;
    }
}

public sealed unsafe class HipLlamaModel : IInferenceProvider     // ≙ LlamaModel.cs:10-175
{
    private readonly ulong model;
    private readonly Tokenizer tokenizer;
    private readonly float[] logits;
    private bool firstInput = true;
    public string ModelName { get; init; }

    public HipLlamaModel(HipBufferManager mgr, Dictionary<string, object> md, List<AbstractComputeCollection> tensors, uint contextSize)
    {
        ModelName = md["general.name"].ToString() ?? "unknown";
        tokenizer = new Tokenizer(md);
        var emb = tensors.First(t => t.Name.Contains("token"));
        var desc = new LlamaDesc
        {
            E = (uint)emb.Shape[0], V = (uint)emb.Shape[1], L = (uint)md["llama.block_count"],
            H = (uint)md["llama.attention.head_count"], Hkv = (uint)md["llama.attention.head_count_kv"],
            D = (uint)md["llama.attention.key_length"],
            F = (uint)tensors.First(t => t.Name.Contains("blk.0.ffn_gate")).Shape[1],
            C = contextSize,
            Eps = (float)(md.First(x => x.Key.Contains("epsilon")).Value),            // LlamaModel.cs:28
            RopeBase = 500000f,                                                        // TransformerBlock.cs:33
            RopeDims = (uint)md["llama.rope.dimension_count"],
            RopeNFreqs = Math.Min(32u, (uint)md["llama.rope.dimension_count"] / 2),    // TransformerBlock.cs:66 (reference-exact);
                                                                                       // use RopeDims/2 for the spec-correct table
            LayerBegin = 0, LayerEnd = (uint)md["llama.block_count"], Flags = 0, MaxBatch = 0,
        };
        Native.Check(Native.nfai_hip_llama_create(mgr.Ctx, in desc, out model));
        foreach (var t in tensors)
        {
            byte[] raw = RawTensorBytes(t);           // on-disk bytes, NOT GetDataRaw (which widens)
            fixed (byte* p = raw)
                Native.Check(Native.nfai_hip_llama_set_tensor(model, t.Name, GgmlTypeOf(t), t.Shape.Length > 1 ? t.Shape[1] : 1,
                                                              t.Shape[0], p));
        }
        Native.Check(Native.nfai_hip_llama_finalize(model));
        logits = new float[desc.V];
    }

    public async IAsyncEnumerable<ChatResponseUpdate> GetStreamingResponseAsync(IEnumerable<ChatMessage> messages,
        ChatOptions? options = null, [EnumeratorCancellation] CancellationToken ct = default)
    {
        var prompt = messages.First(x => x.Role == ChatRole.User).Text;               // LlamaModel.cs:79-80
        var id = Guid.NewGuid().ToString();
        await foreach (var part in RunAsync(prompt, ct))
            yield return new ChatResponseUpdate { ModelId = ModelName, MessageId = id, Contents = [new TextContent(part)] };
    }

    public async IAsyncEnumerable<string> RunAsync(string prompt, [EnumeratorCancellation] CancellationToken ct = default)
    {
        var ids = tokenizer.Tokenize(prompt, addBos: firstInput);                     // LlamaModel.cs:101
        firstInput = false;
        foreach (var t in ids) Step(t);                                               // :103-126
        var tk = SamplingUtils.TopP(logits);                                          // :130
        yield return tokenizer.Detokenize([tk]);
        while (tk != tokenizer.EosTokenId && !ct.IsCancellationRequested)             // :134-173
        {
            Step(tk);
            tk = SamplingUtils.TopP(logits);
            if (tk != tokenizer.EosTokenId) yield return tokenizer.Detokenize([tk]);
        }
        await Task.CompletedTask;
    }

    private void Step(uint token)
    {
        fixed (float* p = logits) Native.Check(Native.nfai_hip_llama_decode_step(model, token, p, out _));
    }

    public void Dispose() => Native.Check(Native.nfai_hip_llama_destroy(model));      // the reference throws NotImplementedException (:70-74)

    private static int GgmlTypeOf(AbstractComputeCollection t) => t.TypeSize == 2 ? (int)GgmlType.F16 : (int)GgmlType.F32;
    private static byte[] RawTensorBytes(AbstractComputeCollection t) => throw new NotImplementedException(
        "read Length*TypeSize bytes at t.offset from the GGUF stream (needs DataStream to be exposed, or re-open ModelOptions.GGUFPath)");
}
