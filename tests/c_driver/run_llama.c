/* run_llama.c — a host that uses libnfai_hip.so through include/nfai_hip.h ONLY (plain C, no Python, no torch, no HIP headers):
 * what the C# NFAI.HIP assembly does by P/Invoke, restated in C for an image without a .NET SDK.
 *
 * It plays LlamaModelFactory.TryCreate + LlamaModel.RunAsync (LlamaModelFactory.cs:24-44, LlamaModel.cs:99-174): create the
 * context, describe the model, hand every GGUF-named tensor over in its on-disk encoding (fp16 matrices, fp32 gains), feed a
 * prompt token by token — and, in a later pass, through nfai_hip_llama_ingest as the drop-in's RunAsync does — then sample greedily
 * and feed back (SamplingUtils.ArgMax, SamplingUtils.cs:43-57).  The same loop
 * runs on the CPU oracle (oracle/libnfai_oracle.so: TEST INFRASTRUCTURE, linked by this test driver only) with the same
 * weights; logits must agree within the end-to-end tolerance of tests/test_gpu_model.py and the tokens must be identical.
 * A second pass uses the device-side greedy loop (nfai_hip_llama_decode_greedy) and must reproduce the tokens.
 *
 * build: gcc -O2 -std=c11 -I include tests/c_driver/run_llama.c -o run_llama -L nfai_amd/csrc -lnfai_hip -L oracle -lnfai_oracle -lm
 * exit code 0 = parity, 1 = mismatch, 2 = API error (message on stderr). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nfai_hip.h"

/* the oracle's C interface (oracle/nfai_oracle.c) */
typedef struct {
    uint32_t E, L, H, Hkv, D, F, V, C;
    float eps, rope_base;
    uint32_t rope_dims, rope_n_freqs, weights_f16;
} orc_llama_desc;
typedef struct orc_llama orc_llama;
orc_llama *orc_llama_create(const orc_llama_desc *d);
void orc_llama_destroy(orc_llama *m);
void orc_llama_set_globals(orc_llama *m, const void *token_embd, const void *output, const float *output_norm);
void orc_llama_set_layer(orc_llama *m, uint32_t l, const float *attn_norm, const void *wq, const void *wk, const void *wv, const void *wo,
                         const float *ffn_norm, const void *wgate, const void *wup, const void *wdown);
int orc_llama_step(orc_llama *m, uint32_t tok, float *logits);
uint32_t orc_argmax(const float *v, uint32_t n);
void orc_llama_reset(orc_llama *m);
uint32_t orc_topp(const float *values, uint32_t n, float temperature, float topP, uint32_t topK, float rand, uint32_t *ids_out, float *probs_out,
                  uint32_t *n_kept_out);
uint16_t orc_float_to_half(float f);

#define CHECK(call)                                                                         \
    do {                                                                                    \
        int32_t _s = (call);                                                                \
        if (_s != 0) {                                                                      \
            fprintf(stderr, "%s failed (%d): %s\n", #call, _s, nfai_hip_last_error());      \
            return 2;                                                                       \
        }                                                                                   \
    } while (0)

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static float rnd(void) /* uniform in [-1, 1), splitmix64 */
{
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)((double)(z >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}

static uint16_t *matrix_f16(uint64_t rows, uint64_t cols, float scale)
{
    uint16_t *w = malloc(rows * cols * 2);
    for (uint64_t i = 0; i < rows * cols; i++) w[i] = orc_float_to_half(scale * rnd());
    return w;
}

static float *gains(uint32_t n)
{
    float *g = malloc(n * 4);
    for (uint32_t i = 0; i < n; i++) g[i] = 1.0f + 0.1f * rnd();
    return g;
}

int main(void)
{
    /* a small Llama-shaped model: 3 blocks, 4 query heads over 2 kv heads of 128, untied lm_head */
    const uint32_t E = 512, L = 3, H = 4, Hkv = 2, D = 128, F = 1024, V = 768, C = 96;
    nfai_device_info info;
    nfai_ctx_t ctx;
    CHECK(nfai_hip_ctx_create(0, &ctx));
    CHECK(nfai_hip_ctx_device_info(ctx, &info));
    printf("device: %s (%s), %u CUs\n", info.name, info.arch, info.compute_units);

    nfai_llama_desc d;
    memset(&d, 0, sizeof d);
    d.E = E; d.L = L; d.H = H; d.Hkv = Hkv; d.D = D; d.F = F; d.V = V; d.C = C;
    d.eps = 1e-5f; d.rope_base = 500000.0f; d.rope_dims = D; d.rope_n_freqs = D / 2;
    d.layer_begin = 0; d.layer_end = L; d.flags = 0;
    d.max_batch = 4;   /* MFMA prefill workspace: the 7 prompt tokens in front of the sampled one go in two chunks (4 + 3) */
    nfai_model_t model;
    CHECK(nfai_hip_llama_create(ctx, &d, &model));

    orc_llama_desc od = {E, L, H, Hkv, D, F, V, C, 1e-5f, 500000.0f, D, D / 2, 1};
    orc_llama *ref = orc_llama_create(&od);
    if (!ref) { fprintf(stderr, "oracle: create failed\n"); return 2; }

    /* tensors by GGUF name, on-disk encoding: ggml type 1 = F16 matrices [ne1 rows][ne0 cols], type 0 = F32 gains */
    uint16_t *emb = matrix_f16(V, E, 0.05f), *out = matrix_f16(V, E, 0.05f);
    float *onorm = gains(E);
    CHECK(nfai_hip_llama_set_tensor(model, "token_embd.weight", 1, V, E, emb));
    CHECK(nfai_hip_llama_set_tensor(model, "output.weight", 1, V, E, out));
    CHECK(nfai_hip_llama_set_tensor(model, "output_norm.weight", 0, 1, E, onorm));
    orc_llama_set_globals(ref, emb, out, onorm);
    for (uint32_t l = 0; l < L; l++) {
        char name[64];
        float *an = gains(E), *fn = gains(E);
        uint16_t *wq = matrix_f16(H * D, E, 0.05f), *wk = matrix_f16(Hkv * D, E, 0.05f), *wv = matrix_f16(Hkv * D, E, 0.05f);
        uint16_t *wo = matrix_f16(E, H * D, 0.05f), *wg = matrix_f16(F, E, 0.05f), *wu = matrix_f16(F, E, 0.05f), *wd = matrix_f16(E, F, 0.05f);
#define SET(fmt, type, rows, cols, ptr)                                       \
    snprintf(name, sizeof name, fmt, l);                                      \
    CHECK(nfai_hip_llama_set_tensor(model, name, type, rows, cols, ptr))
        SET("blk.%u.attn_norm.weight", 0, 1, E, an);
        SET("blk.%u.attn_q.weight", 1, H * D, E, wq);
        SET("blk.%u.attn_k.weight", 1, Hkv * D, E, wk);
        SET("blk.%u.attn_v.weight", 1, Hkv * D, E, wv);
        SET("blk.%u.attn_output.weight", 1, E, H * D, wo);
        SET("blk.%u.ffn_norm.weight", 0, 1, E, fn);
        SET("blk.%u.ffn_gate.weight", 1, F, E, wg);
        SET("blk.%u.ffn_up.weight", 1, F, E, wu);
        SET("blk.%u.ffn_down.weight", 1, E, F, wd);
#undef SET
        orc_llama_set_layer(ref, l, an, wq, wk, wv, wo, fn, wg, wu, wd);
    }
    CHECK(nfai_hip_llama_finalize(model));

    /* LlamaModel.RunAsync: the prompt token by token (LlamaModel.cs:103-126), then sample and feed back (:134-173) */
    const uint32_t prompt[] = {5, 17, 300, 44, 9, 701, 2, 63};
    const uint32_t n_prompt = sizeof prompt / sizeof prompt[0], n_gen = 40;
    float *lg = malloc(V * 4), *lr = malloc(V * 4);
    float *lr_first = malloc(V * 4);   /* the oracle's logits of the last prompt token = of the first sampled step */
    uint32_t tokens[64], am = 0, bad = 0;
    double worst = 0.0;
    for (uint32_t i = 0; i < n_prompt + n_gen; i++) {
        const uint32_t tok = i < n_prompt ? prompt[i] : am;
        CHECK(nfai_hip_llama_decode_step(model, tok, lg, &am));
        if (orc_llama_step(ref, tok, lr) != 0) { fprintf(stderr, "oracle: step failed\n"); return 2; }
        float maxabs = 1.0f, maxd = 0.f;
        for (uint32_t v = 0; v < V; v++) {
            if (fabsf(lr[v]) > maxabs) maxabs = fabsf(lr[v]);
            if (fabsf(lg[v] - lr[v]) > maxd) maxd = fabsf(lg[v] - lr[v]);
        }
        if (maxd / maxabs > worst) worst = maxd / maxabs;
        if (maxd > 5e-4f * maxabs || am != orc_argmax(lr, V)) bad++;   /* tests/test_gpu_model.py: logit_tol */
        if (i >= n_prompt) tokens[i - n_prompt] = tok;
        if (i + 1 == n_prompt) memcpy(lr_first, lr, V * 4);
    }
    uint32_t pos = 0;
    CHECK(nfai_hip_llama_pos(model, &pos));
    printf("step-by-step: %u tokens, worst max|dlogit| / max(1, max|logit|) = %.3g, mismatching steps = %u, position = %u\n",
           n_prompt + n_gen, worst, bad, pos);
    if (bad || pos != n_prompt + n_gen) return 1;

    /* the same generation with the greedy loop on the device (no logits leave the GPU): identical tokens */
    CHECK(nfai_hip_llama_reset(model));
    for (uint32_t i = 0; i + 1 < n_prompt; i++) CHECK(nfai_hip_llama_decode_step(model, prompt[i], NULL, &am));
    uint32_t dev_tokens[64];
    CHECK(nfai_hip_llama_decode_greedy(model, prompt[n_prompt - 1], n_gen, dev_tokens));
    /* tokens_out[k] = the argmax after step k; step 0 feeds the last prompt token, so tokens_out[k] is the token the
     * step-by-step loop fed at generation step k */
    for (uint32_t k = 0; k < n_gen; k++)
        if (dev_tokens[k] != tokens[k]) { fprintf(stderr, "device greedy loop differs at %u: %u vs %u\n", k, dev_tokens[k], tokens[k]); return 1; }
    printf("device greedy loop: %u tokens identical\n", n_gen);

    /* RunAsync as the drop-in runs it (HipLlamaModel.RunAsync, nfai_amd.llama_model.LlamaModel.RunAsync): the prompt tokens in front
     * of the last one in ONE call on the MFMA prefill path (nfai_hip_llama_ingest: K / V rows only, fp16 operands), the last prompt
     * token through the sampled step, then the device-side greedy loop.  Stated fp16 tolerance on the first sampled step's logits
     * (2e-2 * max(1, max|logit|), tests/test_gpu_model.py::test_prefill_mfma_matches_token_by_token), identical greedy tokens. */
    CHECK(nfai_hip_llama_reset(model));
    CHECK(nfai_hip_llama_ingest(model, prompt, n_prompt - 1));
    CHECK(nfai_hip_llama_pos(model, &pos));
    if (pos != n_prompt - 1) { fprintf(stderr, "ingest: position %u, expected %u\n", pos, n_prompt - 1); return 1; }
    CHECK(nfai_hip_llama_decode_step(model, prompt[n_prompt - 1], lg, &am));
    {
        float maxabs = 1.0f, maxd = 0.f;
        for (uint32_t v = 0; v < V; v++) {
            if (fabsf(lr_first[v]) > maxabs) maxabs = fabsf(lr_first[v]);
            if (fabsf(lg[v] - lr_first[v]) > maxd) maxd = fabsf(lg[v] - lr_first[v]);
        }
        printf("prompt through nfai_hip_llama_ingest (MFMA prefill, 4 + 3 tokens): first sampled step max|dlogit| = %.3g (tolerance %.3g)\n",
               maxd, 2e-2f * maxabs);
        if (maxd > 2e-2f * maxabs || am != tokens[0]) { fprintf(stderr, "ingest path: first sampled token %u vs %u\n", am, tokens[0]); return 1; }
    }
    CHECK(nfai_hip_llama_decode_greedy(model, am, n_gen - 1, dev_tokens));
    for (uint32_t k = 0; k + 1 < n_gen; k++)
        if (dev_tokens[k] != tokens[k + 1]) { fprintf(stderr, "ingest path: greedy token %u differs: %u vs %u\n", k + 1, dev_tokens[k], tokens[k + 1]); return 1; }
    printf("ingest path: %u greedy tokens identical to the token-by-token run\n", n_gen);
    /* a prompt that does not fit is refused before anything runs (the reference overruns its cache, MatrixMultiplyShader.cs:248-252) */
    CHECK(nfai_hip_llama_set_pos(model, C - 3));
    if (nfai_hip_llama_ingest(model, prompt, n_prompt - 1) != NFAI_ERR_KV_FULL) { fprintf(stderr, "ingest: expected NFAI_ERR_KV_FULL\n"); return 1; }
    CHECK(nfai_hip_llama_pos(model, &pos));
    if (pos != C - 3) { fprintf(stderr, "ingest: a refused call moved the position to %u\n", pos); return 1; }

    /* the reference's DEFAULT sampler (LlamaModel.cs:128-130: SamplingUtils.TopP on the logits it reads back): the candidates come
     * from the device (nfai_hip_llama_decode_topk), the nucleus cut and the draw are the host's — SamplingUtils.cs:14-31 restated */
    CHECK(nfai_hip_llama_reset(model));
    orc_llama_reset(ref);
    {
        uint32_t tok = prompt[0], ids[40], ids_ref[40], kept = 0, mism = 0;
        float probs[40], probs_ref[40];
        const float rands[6] = {0.03f, 0.31f, 0.5f, 0.77f, 0.94f, 0.999f};
        for (uint32_t i = 0; i < 6; i++) {
            CHECK(nfai_hip_llama_decode_topk(model, tok, 0.5f, 40, ids, probs));
            if (orc_llama_step(ref, tok, lr) != 0) return 2;
            const uint32_t want = orc_topp(lr, V, 0.5f, 0.95f, 40, rands[i], ids_ref, probs_ref, &kept);
            float cumulative = 0.f, total = 0.f, running = 0.f;   /* SamplingUtils.cs:14-31 on the device's candidates */
            uint32_t n = 0, got = 0;
            for (; n < 40;) { cumulative += probs[n]; n++; if (cumulative >= 0.95f) break; }
            { double t = 0.0; for (uint32_t k = 0; k < n; k++) t += probs[k]; total = (float)t; }
            got = ids[n - 1];
            for (uint32_t k = 0; k < n; k++) { running += probs[k] / total; if (rands[i] < running) { got = ids[k]; break; } }
            if (ids[0] != ids_ref[0] || fabsf(probs[0] - probs_ref[0]) > 2e-4f * probs_ref[0] || got != want) mism++;
            tok = want;
        }
        printf("sampling path: 6 steps of decode_topk + the host half of TopP, mismatches against the oracle = %u\n", mism);
        if (mism) return 1;
    }

    /* KV capacity is enforced (the reference writes out of bounds, MatrixMultiplyShader.cs:248-252) */
    CHECK(nfai_hip_llama_set_pos(model, C));
    if (nfai_hip_llama_decode_step(model, 1, NULL, &am) != NFAI_ERR_KV_FULL) { fprintf(stderr, "expected NFAI_ERR_KV_FULL\n"); return 1; }

    CHECK(nfai_hip_llama_destroy(model));
    CHECK(nfai_hip_ctx_destroy(ctx));
    orc_llama_destroy(ref);
    puts("ok");
    return 0;
}
