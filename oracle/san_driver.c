/*
 * san_driver.c — the oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only; never built or run on the GPU box).
 *
 * TEST INFRASTRUCTURE.  The reference runs with the Vulkan validation layer switched on unconditionally
 * (NFAI.Vulkan/VulkanHelper.cs:14-17, errors abort the process, :126-131); this is the build's counterpart for the checker
 * itself: every exported function of nfai_oracle.c is driven once at small, ragged sizes with exactly-sized heap buffers, so an
 * out-of-bounds access, a misaligned or overflowing operation aborts the run.  tests/test_oracle.py compiles (make san_driver)
 * and runs it and expects "san_driver: ok" and a zero exit code.
 */
#include <stdio.h>

#include "nfai_oracle.c"

static uint32_t rng_state = 12345u;
static float frand(void)
{
    rng_state = rng_state * 1664525u + 1013904223u;
    return ((float)(rng_state >> 8) / 8388608.0f) - 1.0f; /* [-1, 1) */
}
static float *fvec(size_t n, float scale)
{
    float *p = (float *)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; i++) p[i] = scale * frand();
    return p;
}

int main(void)
{
    /* fp16 round trip, ragged count */
    {
        float *a = fvec(37, 3.0f), *b = (float *)malloc(37 * 4);
        uint16_t *h = (uint16_t *)malloc(37 * 2);
        orc_narrow_f16(a, h, 37);
        orc_widen_f16(h, b, 37);
        for (int i = 0; i < 37; i++)
            if (fabsf(a[i] - b[i]) > 2e-3f * (1.0f + fabsf(a[i]))) return 2;
        free(a); free(b); free(h);
    }
    /* the 1:1 operators at the edge sizes the GPU tests use (N, K not multiples of anything) */
    {
        const uint32_t N = 7, K = 24, E = 24;
        float *W = fvec((size_t)N * K, 0.1f), *x = fvec(K, 1.0f), *y = (float *)malloc(N * 4), *g = fvec(E, 1.0f), *xn = (float *)malloc(E * 4);
        uint16_t *Wh = (uint16_t *)malloc((size_t)N * K * 2);
        orc_narrow_f16(W, Wh, (size_t)N * K);
        orc_gemv(W, x, y, N, K);
        orc_gemv_f16w(Wh, x, y, N, K);
        orc_rmsnorm(x, g, xn, E, 1e-5f);
        orc_embed(W, N - 1, K, xn);
        orc_silu(x, xn, E); orc_mul(x, g, xn, E); orc_add(x, g, xn, E);
        if (orc_argmax(y, N) >= N) return 3;
        free(W); free(x); free(y); free(g); free(xn); free(Wh);
    }
    /* RoPE + attention chain: H = 4, Hkv = 2, D = 8, S = 5 of C = 6 */
    {
        const uint32_t H = 4, Hkv = 2, D = 8, S = 5, C = 6;
        float *fr = (float *)malloc((D / 2) * 4), *q = fvec(H * D, 1.0f), *qr = (float *)malloc(H * D * 4);
        float *Kc = fvec((size_t)C * Hkv * D, 1.0f), *Vc = fvec((size_t)C * Hkv * D, 1.0f);
        float *s = (float *)malloc((size_t)H * S * 4), *w = (float *)malloc((size_t)H * S * 4), *o = (float *)malloc(H * D * 4);
        orc_rope_freqs(fr, D, 500000.0f, D / 2);
        orc_rope(q, qr, fr, D, H, D, 3);
        orc_attn_scores(qr, Kc, s, H, Hkv, D, S);
        orc_attn_softmax(s, w, H, S, 1e-5f);
        orc_attn_wsum(w, Vc, o, H, Hkv, D, S);
        free(fr); free(q); free(qr); free(Kc); free(Vc); free(s); free(w); free(o);
    }
    /* TopP with k > n, k == n, ties */
    {
        float *v = fvec(50, 3.0f);
        uint32_t ids[40], kept;
        float probs[40];
        v[7] = v[41] = 9.0f;
        if (orc_topp(v, 50, 0.5f, 0.95f, 40, 0.5f, ids, probs, &kept) >= 50 || ids[0] != 7 || kept == 0) return 4;
        if (orc_topp(v, 3, 0.5f, 0.95f, 40, 0.999f, ids, probs, &kept) >= 3) return 4;
        free(v);
    }
    /* K-quant codecs: quantise -> dequantise, two super-blocks, exact-size buffers */
    {
        float *w = fvec(512, 0.05f), *d = (float *)malloc(512 * 4);
        uint8_t *b4 = (uint8_t *)malloc(2 * 144), *b6 = (uint8_t *)malloc(2 * 210);
        orc_quantize_q4k(w, b4, 512); orc_dequant_q4k(b4, d, 512);
        for (int i = 0; i < 512; i++) if (fabsf(d[i] - w[i]) > 0.02f) return 5;
        orc_quantize_q6k(w, b6, 512); orc_dequant_q6k(b6, d, 512);
        for (int i = 0; i < 512; i++) if (fabsf(d[i] - w[i]) > 0.005f) return 5;
        free(w); free(d); free(b4); free(b6);
    }
    /* whole model, fp32 and fp16 weights, every position of the cache (C steps), then one step too many must be refused */
    for (int f16 = 0; f16 < 2; f16++) {
        orc_llama_desc d = {.E = 32, .L = 2, .H = 4, .Hkv = 2, .D = 8, .F = 48, .V = 40, .C = 5, .eps = 1e-5f, .rope_base = 500000.0f,
                            .rope_dims = 8, .rope_n_freqs = 4, .weights_f16 = (uint32_t)f16};
        orc_llama *m = orc_llama_create(&d);
        const uint32_t HD = d.H * d.D, KD = d.Hkv * d.D;
        void *mats[2 + 2 * 7];
        float *norms[1 + 2 * 2];
        int nm = 0, nn = 0;
#define MAT(r, c) ({ float *t_ = fvec((size_t)(r) * (c), 0.08f); void *o_ = t_; \
                     if (f16) { uint16_t *h_ = (uint16_t *)malloc((size_t)(r) * (c) * 2); orc_narrow_f16(t_, h_, (size_t)(r) * (c)); free(t_); o_ = h_; } \
                     mats[nm++] = o_; o_; })
        void *emb = MAT(d.V, d.E);
        norms[nn++] = fvec(d.E, 1.0f);
        orc_llama_set_globals(m, emb, NULL, norms[0]);
        for (uint32_t l = 0; l < d.L; l++) {
            float *an = fvec(d.E, 1.0f), *fn = fvec(d.E, 1.0f);
            norms[nn++] = an; norms[nn++] = fn;
            void *wq = MAT(HD, d.E), *wk = MAT(KD, d.E), *wv = MAT(KD, d.E), *wo = MAT(d.E, HD), *wg = MAT(d.F, d.E), *wu = MAT(d.F, d.E), *wd = MAT(d.E, d.F);
            orc_llama_set_layer(m, l, an, wq, wk, wv, wo, fn, wg, wu, wd);
        }
        float *logits = (float *)malloc(d.V * 4);
        uint32_t tok = 3;
        for (uint32_t p = 0; p < d.C; p++) {
            if (orc_llama_step(m, tok, logits) != 0) return 6;
            tok = orc_argmax(logits, d.V);
        }
        if (orc_llama_step(m, tok, logits) == 0) return 7; /* KV capacity is a hard error */
        orc_llama_reset(m);
        if (orc_llama_step(m, 1, logits) != 0 || orc_llama_pos(m) != 1) return 8;
        free(logits);
        orc_llama_destroy(m);
        for (int i = 0; i < nm; i++) free(mats[i]);
        for (int i = 0; i < nn; i++) free(norms[i]);
    }
    printf("san_driver: ok\n");
    return 0;
}
