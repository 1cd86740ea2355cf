/*
 * lnb_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see lnb_oracle.h).
 *
 * Plain-C restatement of adalkiran/llama-nuts-and-bolts' LlamaTransformer.Forward
 * arithmetic.  Build with:  gcc -O2 -fopenmp -ffp-contract=off -fno-fast-math
 * (-ffp-contract=off reproduces the amd64 Go compiler, which never fuses a*b+c;
 * bf16*bf16 products are exact in f32 anyway, so this only matters on underflow).
 *
 * Rounding contract: every bf16 store is a TRUNCATION
 * (bits>>16, src/dtype/bfloat16.go:31-33), every matmul accumulates in f32
 * strictly sequentially in ascending k (operations_lineartransform.go:46-65),
 * softmax is f64 without max subtraction (operations_impl.go:492-508), argmax is
 * first-max-wins (operations_impl.go:529-541).
 *
 * Parallelism: OpenMP over OUTPUT elements only (never over k), mirroring the
 * reference's goroutine-per-output fan-out (operations_lineartransform.go:119-130).
 */
#include "lnb_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <omp.h>
#include <time.h>

static __thread char g_err[512];
const char* orc_last_error(void) { return g_err; }
static int fail(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); return -1;
}

/* ------------------------------------------------------------------ dtype */
static inline float wide(uint16_t b) { union { uint32_t u; float f; } v; v.u = (uint32_t)b << 16; return v.f; }
static inline uint16_t trunc16(float f) { union { uint32_t u; float f; } v; v.f = f; return (uint16_t)(v.u >> 16); }
uint16_t orc_f32_to_bf16(float f) { return trunc16(f); }   /* bfloat16.go:31-33 */
float orc_bf16_to_f32(uint16_t b) { return wide(b); }      /* bfloat16.go:19-21 */

/* ------------------------------------------------------- synthetic weights */
static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
/* Irwin-Hall(8) of 16-bit uniforms: integer-exact, no transcendental => identical on host and device */
static inline float synth_gauss(uint64_t seed, uint32_t tensor_id, uint64_t idx) {
    uint64_t base = splitmix64(seed + 0x9E3779B97F4A7C15ULL * (uint64_t)(tensor_id + 1u));
    uint64_t a = splitmix64(base ^ idx);
    uint64_t b = splitmix64(a);
    int32_t s = 0;
    for (int i = 0; i < 4; i++) { s += (int32_t)((a >> (16 * i)) & 0xFFFF); s += (int32_t)((b >> (16 * i)) & 0xFFFF); }
    s -= 262140;                                   /* 8 * 32767.5 */
    return (float)s * (1.0f / 53510.0f);           /* std of the sum = 65536*sqrt(8/12) ~= 53510 */
}
uint16_t orc_synth_bf16(uint64_t seed, uint32_t tensor_id, uint64_t idx, int kind, float sigma) {
    float g = synth_gauss(seed, tensor_id, idx);
    float v = kind == 1 ? fmaf(0.1f, g, 1.0f) : sigma * g;
    return trunc16(v);
}
int32_t orc_synth_token(uint64_t seed, uint64_t i, int32_t vocab) {
    return (int32_t)(splitmix64(seed ^ (i * 0x9E3779B97F4A7C15ULL)) % (uint64_t)vocab);
}
void orc_synth_fill(uint16_t* dst, uint64_t n, uint64_t seed, uint32_t tensor_id, int kind, float sigma) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) dst[i] = orc_synth_bf16(seed, tensor_id, (uint64_t)i, kind, sigma);
}

/* ------------------------------------------------------------------ linear */
/* operations_lineartransform.go:46-65: valDstF32 += val1F32 * val2F32, k ascending, then
   dstF32.ToBFloat16() (:205).  Row tile of MR inputs x NR outputs keeps each (m,n) chain intact. */
#define MR 4
#define NR 4
static void linear_core(const uint16_t* x, const uint16_t* w, uint16_t* y, int rows, int N, int K, int nthreads) {
    float* xf = (float*)malloc((size_t)rows * K * sizeof(float));
    for (int64_t i = 0; i < (int64_t)rows * K; i++) xf[i] = wide(x[i]);
    if (nthreads < 1) nthreads = 1;
    if (rows == 1) {   /* decode: 8 independent output chains interleaved for ILP; each chain is still k-sequential */
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
        for (int nb = 0; nb < (N + 7) / 8; nb++) {
            int n0 = nb * 8, nn = N - n0 < 8 ? N - n0 : 8;
            float acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
            const uint16_t* wr[8];
            for (int b = 0; b < 8; b++) wr[b] = w + (size_t)(n0 + (b < nn ? b : 0)) * K;
            for (int k = 0; k < K; k++) {
                float xv = xf[k];
                for (int b = 0; b < 8; b++) { float p = xv * wide(wr[b][k]); acc[b] += p; }
            }
            for (int b = 0; b < nn; b++) y[n0 + b] = trunc16(acc[b]);
        }
        free(xf);
        return;
    }
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
    for (int nb = 0; nb < (N + NR - 1) / NR; nb++) {
        int n0 = nb * NR, nn = N - n0 < NR ? N - n0 : NR;
        for (int m0 = 0; m0 < rows; m0 += MR) {
            int mm = rows - m0 < MR ? rows - m0 : MR;
            float acc[MR][NR];
            for (int a = 0; a < MR; a++) for (int b = 0; b < NR; b++) acc[a][b] = 0.0f;
            if (mm == MR && nn == NR) {
                const uint16_t* w0 = w + (size_t)(n0 + 0) * K; const uint16_t* w1 = w + (size_t)(n0 + 1) * K;
                const uint16_t* w2 = w + (size_t)(n0 + 2) * K; const uint16_t* w3 = w + (size_t)(n0 + 3) * K;
                const float* x0 = xf + (size_t)(m0 + 0) * K; const float* x1 = xf + (size_t)(m0 + 1) * K;
                const float* x2 = xf + (size_t)(m0 + 2) * K; const float* x3 = xf + (size_t)(m0 + 3) * K;
                for (int k = 0; k < K; k++) {
                    float wv[NR] = { wide(w0[k]), wide(w1[k]), wide(w2[k]), wide(w3[k]) };
                    float xv[MR] = { x0[k], x1[k], x2[k], x3[k] };
                    for (int a = 0; a < MR; a++) for (int b = 0; b < NR; b++) { float p = xv[a] * wv[b]; acc[a][b] += p; }
                }
            } else {
                for (int a = 0; a < mm; a++) for (int b = 0; b < nn; b++) {
                    const uint16_t* wr = w + (size_t)(n0 + b) * K; const float* xr = xf + (size_t)(m0 + a) * K;
                    float s = 0.0f;
                    for (int k = 0; k < K; k++) { float p = xr[k] * wide(wr[k]); s += p; }
                    acc[a][b] = s;
                }
            }
            for (int a = 0; a < mm; a++) for (int b = 0; b < nn; b++) y[(size_t)(m0 + a) * N + n0 + b] = trunc16(acc[a][b]);
        }
    }
    free(xf);
}
void orc_linear_bf16(const uint16_t* x, const uint16_t* w, uint16_t* y, int rows, int n_out, int k_in, int nthreads) {
    linear_core(x, w, y, rows, n_out, k_in, nthreads);
}
/* operations_lineartransform.go:72-103 + :205 */
void orc_linear_f32(const float* x, const float* w, float* y_f32, uint16_t* y_bf16, int rows, int n_out, int k_in) {
    for (int m = 0; m < rows; m++) for (int n = 0; n < n_out; n++) {
        float s = 0.0f;
        for (int k = 0; k < k_in; k++) { float p = x[(size_t)m * k_in + k] * w[(size_t)n * k_in + k]; s += p; }
        if (y_f32) y_f32[(size_t)m * n_out + n] = s;
        if (y_bf16) y_bf16[(size_t)m * n_out + n] = trunc16(s);
    }
}
/* operations_matmul.go:37-55 (other read with stride n), :180 ToBFloat16 */
void orc_matmul_bf16(const uint16_t* a, const uint16_t* b, uint16_t* c, int batch, int m, int k, int n) {
    for (int bi = 0; bi < batch; bi++) for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) {
        float s = 0.0f;
        for (int kk = 0; kk < k; kk++) {
            float p = wide(a[((size_t)bi * m + i) * k + kk]) * wide(b[((size_t)bi * k + kk) * n + j]);
            s += p;
        }
        c[((size_t)bi * m + i) * n + j] = trunc16(s);
    }
}

/* -------------------------------------------------------------- small ops */
int orc_arange_bf16(int start, int end, int step, uint16_t* out) {   /* operations_impl.go:11-24 */
    if (start >= end) return -1;
    int i = 0; for (int v = start; v < end; v += step) out[i++] = trunc16((float)v); return i;
}
int orc_arange_f32(int start, int end, int step, float* out) {
    if (start >= end) return -1;
    int i = 0; for (int v = start; v < end; v += step) out[i++] = (float)v; return i;
}
void orc_outer_bf16(const uint16_t* v1, int n1, const uint16_t* v2, int n2, uint16_t* out) {   /* :26-53 */
    for (int i = 0; i < n1; i++) for (int j = 0; j < n2; j++) out[(size_t)i * n2 + j] = trunc16(wide(v1[i]) * wide(v2[j]));
}
void orc_polar_f32(const float* abs_, const float* angle, float* o, int n) {   /* :100-140 */
    for (int i = 0; i < n; i++) {
        double ab = (double)abs_[i], an = (double)angle[i];
        o[2 * i] = (float)(ab * cos(an)); o[2 * i + 1] = (float)(ab * sin(an));
    }
}
void orc_polar_bf16(const uint16_t* abs_, const uint16_t* angle, float* o, int n) {
    for (int i = 0; i < n; i++) {
        double ab = (double)wide(abs_[i]), an = (double)wide(angle[i]);
        o[2 * i] = (float)(ab * cos(an)); o[2 * i + 1] = (float)(ab * sin(an));
    }
}
void orc_triu_bf16(const uint16_t* in, uint16_t* out, int rows, int cols, int diagonal) {   /* :175-195 */
    for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++)
        out[(size_t)i * cols + j] = (j - i >= diagonal) ? in[(size_t)i * cols + j] : 0;
}
void orc_pow_bf16(const uint16_t* in, float* out, int n, double power) {   /* :197-217 */
    for (int i = 0; i < n; i++) out[i] = (float)pow((double)wide(in[i]), power);
}
void orc_mean_f32(const float* in, float* out, int groups, int last) {   /* :219-253 */
    for (int g = 0; g < groups; g++) {
        float s = 0.0f; for (int i = 0; i < last; i++) s += in[(size_t)g * last + i];
        out[g] = s / (float)last;
    }
}
/* Which float64 exp the oracle calls (test infrastructure: tests/test_exp_implementations.py).  0 (default): the host libm's.  1: Go's PORTABLE math.Exp restated
 * (src/math/exp.go = FreeBSD msun e_exp.c: two-part ln2 reduction, degree-5 polynomial in r^2, ldexp) -- the reference's softmax and SiLU table call math.Exp
 * (operations_impl.go:498/506, activations.go:24); the two differ by one ulp on ~0.75 % of the bf16 inputs, and the switch exists to show that NOT ONE logit bit of a whole
 * model run depends on which one is used (compiled with -ffp-contract=off like the rest: no fused operations). */
static int g_exp_impl = 0;
static double exp_go(double x) {
    static const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10, Log2e = 1.44269504088896338700e+00,
        Overflow = 7.09782712893383973096e+02, Underflow = -7.45133219101941108420e+02, NearZero = 1.0 / (1 << 28),
        P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x || x == INFINITY) return x;
    if (x == -INFINITY) return 0.0;
    if (x > Overflow) return INFINITY;
    if (x < Underflow) return 0.0;
    if (-NearZero < x && x < NearZero) return 1.0 + x;
    int k = 0;
    if (x < 0) k = (int)(Log2e * x - 0.5); else if (x > 0) k = (int)(Log2e * x + 0.5);
    const double hi = x - (double)k * Ln2Hi, lo = (double)k * Ln2Lo;
    const double r = hi - lo, t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    return ldexp(y, k);
}
static inline double orc_exp(double x) { return g_exp_impl ? exp_go(x) : exp(x); }
static int g_silu_init;                                     /* (defined below; the table is rebuilt after a switch) */
void orc_set_exp_impl(int which) { g_exp_impl = which ? 1 : 0; g_silu_init = 0; }
double orc_exp_f64(double x) { return orc_exp(x); }
void orc_softmax_f32(const float* in, float* out, int rows, int cols) {   /* :478-511 */
    for (int r = 0; r < rows; r++) {
        double z = 0.0;
        for (int j = 0; j < cols; j++) z += orc_exp((double)in[(size_t)r * cols + j]);
        for (int j = 0; j < cols; j++) out[(size_t)r * cols + j] = (float)(orc_exp((double)in[(size_t)r * cols + j]) / z);
    }
}
int32_t orc_argmax_f32(const float* in, int n) {   /* :529-541 */
    float mx = -3.40282346638528859811704183484516925440e+38f; int32_t mi = -1;
    for (int i = 0; i < n; i++) if (mx < in[i]) { mx = in[i]; mi = i; }
    return mi;
}
static float g_silu[1 << 16];
const float* orc_silu_table(void) {   /* activations.go:15-25 */
    if (!g_silu_init) {
#pragma omp critical
        {
            if (!g_silu_init) {
                for (int i = 0; i < (1 << 16); i++) { double x = (double)wide((uint16_t)i); g_silu[i] = (float)(x / (1.0 + orc_exp(-x))); }
                g_silu_init = 1;
            }
        }
    }
    return g_silu;
}

/* ---------------------------------------------------------------- RMSNorm */
/* llamatransformer.go:633-660 -> Pow(x,2) impl:197-217; Mean impl:236-251; AddScalar :262-267;
   RSqrt :298-301; MultiplyElementwise(x,h) -> bf16 (trunc); MultiplyElementwise(h, weights) -> bf16 */
void orc_rmsnorm_bf16(const uint16_t* x, const uint16_t* w, uint16_t* y, int rows, int dim, float eps, uint16_t* pre) {
    for (int r = 0; r < rows; r++) {
        const uint16_t* xr = x + (size_t)r * dim;
        float sum = 0.0f;
        for (int k = 0; k < dim; k++) { double xd = (double)wide(xr[k]); float p = (float)(xd * xd); sum += p; }
        float mean = sum / (float)dim;
        mean = mean + eps;
        float rs = (float)(1.0 / sqrt((double)mean));
        for (int k = 0; k < dim; k++) {
            uint16_t h = trunc16(wide(xr[k]) * rs);
            if (pre) pre[(size_t)r * dim + k] = h;
            y[(size_t)r * dim + k] = trunc16(wide(h) * wide(w[k]));
        }
    }
}

/* ------------------------------------------------------------------- RoPE */
/* llamatransformer.go:662-692 (all f32 arithmetic), :694-751 */
void orc_rope_freqs(int head_dim, double theta, int use_scaled, uint16_t* freqs) {
    int n = head_dim / 2;
    float dimf = (float)head_dim;
    for (int i = 0; i < n; i++) {
        float val = wide(trunc16((float)(2 * i)));                       /* ARange(0,dim,2,BF16) */
        freqs[i] = trunc16((float)(1.0 / pow(theta, (double)(val / dimf))));
    }
    if (use_scaled) {
        const float scaleFactor = 8.0f, lowFreqFactor = 1.0f, highFreqFactor = 4.0f, oldContextLen = 8192.0f;
        const float lowFreqWavelen = oldContextLen / lowFreqFactor, highFreqWavelen = oldContextLen / highFreqFactor;
        for (int i = 0; i < n; i++) {
            float freq = wide(freqs[i]), nf;
            float wavelen = (float)(2 * M_PI) / freq;
            if (wavelen < highFreqWavelen) nf = freq;
            else if (wavelen > lowFreqWavelen) nf = freq / scaleFactor;
            else {
                float smooth = (oldContextLen / wavelen - lowFreqFactor) / (highFreqFactor - lowFreqFactor);
                float t1 = (1 - smooth) * freq; t1 = t1 / scaleFactor;
                float t2 = smooth * freq;
                nf = t1 + t2;
            }
            freqs[i] = trunc16(nf);
        }
    }
}
void orc_rope_table(int head_dim, int rows, double theta, int use_scaled, float* cis, uint16_t* angles_opt) {
    int n = head_dim / 2;
    uint16_t* freqs = (uint16_t*)malloc(n * sizeof(uint16_t));
    orc_rope_freqs(head_dim, theta, use_scaled, freqs);
    for (int p = 0; p < rows; p++) {
        float t = wide(trunc16((float)p));                               /* ARange(0,end,1,BF16): p>=256 quantised */
        for (int i = 0; i < n; i++) {
            uint16_t a16 = trunc16(t * wide(freqs[i]));                  /* Outer, impl:26-53 */
            if (angles_opt) angles_opt[(size_t)p * n + i] = a16;
            double a = (double)wide(a16);
            double one = (double)wide(trunc16(1.0f));                    /* OnesLike(freqs) bf16 */
            cis[((size_t)p * n + i) * 2 + 0] = (float)(one * cos(a));   /* Polar impl:131-133 */
            cis[((size_t)p * n + i) * 2 + 1] = (float)(one * sin(a));
        }
    }
    free(freqs);
}
/* llamatransformer.go:753-790; complex64 multiply is evaluated in float64 by the Go compiler and narrowed
   (cmd/compile ssagen: "Compute in Float64 to minimize cancellation error"); SURVEY.md Appendix A N6 */
void orc_rope_apply(uint16_t* x, int S, int n_heads, int head_dim, const float* cis_rows) {
    int n = head_dim / 2;
    for (int s = 0; s < S; s++) for (int h = 0; h < n_heads; h++) for (int i = 0; i < n; i++) {
        uint16_t* px = x + ((size_t)s * n_heads + h) * head_dim + 2 * i;
        double a = (double)wide(px[0]), b = (double)wide(px[1]);
        double c = (double)cis_rows[((size_t)s * n + i) * 2], d = (double)cis_rows[((size_t)s * n + i) * 2 + 1];
        float re = (float)(a * c - b * d);
        float im = (float)(a * d + b * c);
        px[0] = trunc16(re); px[1] = trunc16(im);
    }
}

/* ------------------------------------------------------------------ model */
#define MAX_TENSORS 4096
typedef struct { char name[96]; uint16_t* data; int64_t nelem; int rows, cols; } orc_tensor;
struct orc_model {
    orc_args a; int head_dim, n_rep, ffn_hidden;
    orc_tensor* t; int nt;
    float* cis; int cis_rows;
    int finalized;
};
struct orc_ctx {
    orc_model* m; int seq_len; int nthreads;
    uint16_t** ck; uint16_t** cv;
    orc_dump_fn dump; void* dump_user;
};

int orc_ffn_hidden_dim(const orc_args* a) {   /* llamatransformer.go:569-577 */
    int h = 4 * a->dim;
    h = (int)(2 * h / 3);
    if (a->ffn_dim_multiplier > -1) h = (int)(a->ffn_dim_multiplier * (double)h);
    h = a->multiple_of * ((h + a->multiple_of - 1) / a->multiple_of);
    return h;
}
static orc_tensor* find_tensor(orc_model* m, const char* name) {
    for (int i = 0; i < m->nt; i++) if (!strcmp(m->t[i].name, name)) return &m->t[i];
    return NULL;
}
static void add_tensor(orc_model* m, const char* name, int rows, int cols) {
    orc_tensor* t = &m->t[m->nt++];
    snprintf(t->name, sizeof t->name, "%s", name); t->rows = rows; t->cols = cols;
    t->nelem = (int64_t)rows * cols; t->data = (uint16_t*)calloc((size_t)t->nelem, sizeof(uint16_t));
}
orc_model* orc_model_create(const orc_args* a) {
    orc_model* m = (orc_model*)calloc(1, sizeof *m);
    m->a = *a;
    if (m->a.n_kv_heads < 0) m->a.n_kv_heads = m->a.n_heads;          /* llamatransformer.go:73-75 */
    m->n_rep = m->a.n_heads / m->a.n_kv_heads;
    m->head_dim = m->a.dim / m->a.n_heads;
    if (m->a.rope_theta <= 0) m->a.rope_theta = 500000.0;               /* :80-82 */
    m->ffn_hidden = orc_ffn_hidden_dim(&m->a);
    m->t = (orc_tensor*)calloc(MAX_TENSORS, sizeof(orc_tensor));
    int dim = m->a.dim, kvd = m->a.n_kv_heads * m->head_dim, qd = m->a.n_heads * m->head_dim;
    char nm[96];
    add_tensor(m, "tok_embeddings.weight", m->a.vocab_size, dim);
    for (int l = 0; l < m->a.n_layers; l++) {
        snprintf(nm, sizeof nm, "layers.%d.attention_norm.weight", l); add_tensor(m, nm, 1, dim);
        snprintf(nm, sizeof nm, "layers.%d.attention.wq.weight", l); add_tensor(m, nm, qd, dim);
        snprintf(nm, sizeof nm, "layers.%d.attention.wk.weight", l); add_tensor(m, nm, kvd, dim);
        snprintf(nm, sizeof nm, "layers.%d.attention.wv.weight", l); add_tensor(m, nm, kvd, dim);
        snprintf(nm, sizeof nm, "layers.%d.attention.wo.weight", l); add_tensor(m, nm, qd, dim);
        snprintf(nm, sizeof nm, "layers.%d.ffn_norm.weight", l); add_tensor(m, nm, 1, dim);
        snprintf(nm, sizeof nm, "layers.%d.feed_forward.w1.weight", l); add_tensor(m, nm, m->ffn_hidden, dim);
        snprintf(nm, sizeof nm, "layers.%d.feed_forward.w2.weight", l); add_tensor(m, nm, dim, m->ffn_hidden);
        snprintf(nm, sizeof nm, "layers.%d.feed_forward.w3.weight", l); add_tensor(m, nm, m->ffn_hidden, dim);
    }
    add_tensor(m, "norm.weight", 1, dim);
    add_tensor(m, "output.weight", m->a.vocab_size, dim);
    return m;
}
void orc_model_destroy(orc_model* m) {
    if (!m) return;
    for (int i = 0; i < m->nt; i++) free(m->t[i].data);
    free(m->t); free(m->cis); free(m);
}
int orc_model_set_tensor(orc_model* m, const char* name, const uint16_t* data, int64_t nelem) {
    orc_tensor* t = find_tensor(m, name);
    if (!t) return fail("unknown tensor %s", name);
    if (t->nelem != nelem) return fail("tensor %s: expected %lld elements, got %lld", name, (long long)t->nelem, (long long)nelem);
    memcpy(t->data, data, (size_t)nelem * 2); return 0;
}
const uint16_t* orc_model_get_tensor(orc_model* m, const char* name, int64_t* nelem) {
    orc_tensor* t = find_tensor(m, name); if (!t) return NULL; if (nelem) *nelem = t->nelem; return t->data;
}
/* tensor ids of the synthetic generator: 0 tok_embd, 1 norm, 2 output,
   16*(l+1)+{0 attn_norm,1 wq,2 wk,3 wv,4 wo,5 ffn_norm,6 w1,7 w2,8 w3} */
static int synth_id(const char* name, int* kind) {
    int l; char rest[64];
    *kind = 0;
    if (!strcmp(name, "tok_embeddings.weight")) return 0;
    if (!strcmp(name, "norm.weight")) { *kind = 1; return 1; }
    if (!strcmp(name, "output.weight")) return 2;
    if (sscanf(name, "layers.%d.%63s", &l, rest) == 2) {
        int base = 16 * (l + 1);
        if (!strcmp(rest, "attention_norm.weight")) { *kind = 1; return base + 0; }
        if (!strcmp(rest, "attention.wq.weight")) return base + 1;
        if (!strcmp(rest, "attention.wk.weight")) return base + 2;
        if (!strcmp(rest, "attention.wv.weight")) return base + 3;
        if (!strcmp(rest, "attention.wo.weight")) return base + 4;
        if (!strcmp(rest, "ffn_norm.weight")) { *kind = 1; return base + 5; }
        if (!strcmp(rest, "feed_forward.w1.weight")) return base + 6;
        if (!strcmp(rest, "feed_forward.w2.weight")) return base + 7;
        if (!strcmp(rest, "feed_forward.w3.weight")) return base + 8;
    }
    return -1;
}
void orc_model_fill_synthetic(orc_model* m, uint64_t seed, int nthreads) {
    if (nthreads > 0) omp_set_num_threads(nthreads);
    for (int i = 0; i < m->nt; i++) {
        int kind, id = synth_id(m->t[i].name, &kind);
        orc_synth_fill(m->t[i].data, (uint64_t)m->t[i].nelem, seed, (uint32_t)id, kind, 0.02f);
    }
}
int orc_model_finalize(orc_model* m) {
    /* llamatransformer.go:109 : precomputeFreqsCis(dim/n_heads, MaxSequenceLength*2, theta, scaled) */
    m->cis_rows = m->a.max_seq_len * 2;
    free(m->cis);
    m->cis = (float*)malloc((size_t)m->cis_rows * (m->head_dim / 2) * 2 * sizeof(float));
    orc_rope_table(m->head_dim, m->cis_rows, m->a.rope_theta, m->a.use_scaled_rope, m->cis, NULL);
    orc_silu_table();
    m->finalized = 1; return 0;
}
const float* orc_model_rope_table(orc_model* m, int* rows) { if (rows) *rows = m->cis_rows; return m->cis; }

orc_ctx* orc_ctx_create(orc_model* m, int seq_len) {   /* inferencecontext.go:17-46 */
    orc_ctx* c = (orc_ctx*)calloc(1, sizeof *c);
    c->m = m; c->seq_len = seq_len > 0 ? seq_len : m->a.max_seq_len; c->nthreads = omp_get_max_threads();
    c->ck = (uint16_t**)calloc(m->a.n_layers, sizeof(uint16_t*)); c->cv = (uint16_t**)calloc(m->a.n_layers, sizeof(uint16_t*));
    size_t n = (size_t)c->seq_len * m->a.n_kv_heads * m->head_dim;
    for (int l = 0; l < m->a.n_layers; l++) { c->ck[l] = (uint16_t*)calloc(n, 2); c->cv[l] = (uint16_t*)calloc(n, 2); }
    return c;
}
void orc_ctx_destroy(orc_ctx* c) {
    if (!c) return;
    for (int l = 0; l < c->m->a.n_layers; l++) { free(c->ck[l]); free(c->cv[l]); }
    free(c->ck); free(c->cv); free(c);
}
void orc_ctx_set_threads(orc_ctx* c, int n) { c->nthreads = n > 0 ? n : 1; }
const uint16_t* orc_ctx_cache(orc_ctx* c, int layer, int which) { return which ? c->cv[layer] : c->ck[layer]; }
void orc_ctx_set_dump(orc_ctx* c, orc_dump_fn fn, void* user) { c->dump = fn; c->dump_user = user; }

#define DUMP(c, stage, layer, dtype, ptr, ...) do { if ((c)->dump) { int64_t shp[] = { __VA_ARGS__ }; \
    (c)->dump((c)->dump_user, stage, layer, dtype, ptr, shp, (int)(sizeof shp / sizeof shp[0])); } } while (0)

static const uint16_t* layer_tensor(orc_model* m, int l, const char* suffix) {
    char nm[96]; snprintf(nm, sizeof nm, "layers.%d.%s", l, suffix);
    orc_tensor* t = find_tensor(m, nm); return t ? t->data : NULL;
}

/* LlamaAttention.Forward, llamatransformer.go:289-527.  x = normalised input [S,dim]; out [S,dim] */
static int attention_forward(orc_ctx* c, int layer, const uint16_t* x, int S, int start_pos, uint16_t* out) {
    orc_model* m = c->m; const orc_args* a = &m->a;
    int dim = a->dim, H = a->n_heads, KVH = a->n_kv_heads, hd = m->head_dim, nrep = m->n_rep;
    int qd = H * hd, kvd = KVH * hd, T = start_pos + S;
    uint16_t* xq = (uint16_t*)malloc((size_t)S * qd * 2);
    uint16_t* xk = (uint16_t*)malloc((size_t)S * kvd * 2);
    uint16_t* xv = (uint16_t*)malloc((size_t)S * kvd * 2);
    /* :297-366 three LinearTransformations */
    linear_core(x, layer_tensor(m, layer, "attention.wq.weight"), xq, S, qd, dim, c->nthreads);
    linear_core(x, layer_tensor(m, layer, "attention.wk.weight"), xk, S, kvd, dim, c->nthreads);
    linear_core(x, layer_tensor(m, layer, "attention.wv.weight"), xv, S, kvd, dim, c->nthreads);
    DUMP(c, "xq", layer, 0, xq, S, qd); DUMP(c, "xk", layer, 0, xk, S, kvd); DUMP(c, "xv", layer, 0, xv, S, kvd);
    /* :392 RoPE with freqs_cis rows start_pos..start_pos+S-1 (slice checked in orc_forward) */
    const float* cis = m->cis + (size_t)start_pos * (hd / 2) * 2;
    orc_rope_apply(xq, S, H, hd, cis);
    orc_rope_apply(xk, S, KVH, hd, cis);
    DUMP(c, "xq_rope", layer, 0, xq, S, H, hd); DUMP(c, "xk_rope", layer, 0, xk, S, KVH, hd);
    /* :402-403 cache update */
    memcpy(c->ck[layer] + (size_t)start_pos * kvd, xk, (size_t)S * kvd * 2);
    memcpy(c->cv[layer] + (size_t)start_pos * kvd, xv, (size_t)S * kvd * 2);
    const uint16_t* K = c->ck[layer]; const uint16_t* V = c->cv[layer];      /* rows 0..T-1, :409-416 */
    /* :464 divisor = bf16(float32(sqrt(head_dim))) */
    float divisor = wide(trunc16((float)sqrt((double)hd)));
    uint16_t neg_inf = trunc16(-INFINITY);
    uint16_t* scores_dump = c->dump ? (uint16_t*)malloc((size_t)H * S * T * 2) : NULL;
    uint16_t* raw_dump = c->dump ? (uint16_t*)malloc((size_t)H * S * T * 2) : NULL;       /* scores after the division (:464), before the mask */
    uint16_t* masked_dump = c->dump ? (uint16_t*)malloc((size_t)H * S * T * 2) : NULL;    /* ... after Add(scores, mask) (:469-473) */
    uint16_t* att = (uint16_t*)malloc((size_t)S * qd * 2);                    /* [S, H*hd] after transpose+reshape :508-514 */
#pragma omp parallel for collapse(2) schedule(dynamic, 1) num_threads(c->nthreads)
    for (int h = 0; h < H; h++) for (int i = 0; i < S; i++) {
        int kvh = h / nrep;                                                    /* attentionRepeatKV :529-559 */
        uint16_t* s16 = (uint16_t*)malloc((size_t)T * 2);
        const uint16_t* q = xq + ((size_t)i * H + h) * hd;
        for (int j = 0; j < T; j++) {
            const uint16_t* kr = K + ((size_t)j * KVH + kvh) * hd;
            float acc = 0.0f;                                                  /* MatMul :459, operations_matmul.go:37-55 */
            for (int d = 0; d < hd; d++) { float p = wide(q[d]) * wide(kr[d]); acc += p; }
            uint16_t s = trunc16(acc);
            s = trunc16(wide(s) / divisor);                                    /* DivToScalar :464 */
            if (raw_dump) raw_dump[((size_t)h * S + i) * T + j] = s;
            if (S > 1) {                                                       /* Add(scores, mask) :469-473, modulo broadcast */
                int jj = j % S;
                uint16_t mk = (jj - i >= 1) ? neg_inf : 0;
                s = trunc16(wide(s) + wide(mk));
            }
            s16[j] = s;
            if (masked_dump) masked_dump[((size_t)h * S + i) * T + j] = s;
        }
        /* :484-495 ToFloat32 -> Softmax (f64) -> ToBFloat16 */
        double z = 0.0;
        for (int j = 0; j < T; j++) z += orc_exp((double)wide(s16[j]));
        for (int j = 0; j < T; j++) s16[j] = trunc16((float)(orc_exp((double)wide(s16[j])) / z));
        if (scores_dump) memcpy(scores_dump + ((size_t)h * S + i) * T, s16, (size_t)T * 2);
        /* :504 MatMul(scores, values) */
        for (int d = 0; d < hd; d++) {
            float acc = 0.0f;
            for (int j = 0; j < T; j++) { float p = wide(s16[j]) * wide(V[((size_t)j * KVH + kvh) * hd + d]); acc += p; }
            att[(size_t)i * qd + (size_t)h * hd + d] = trunc16(acc);
        }
        free(s16);
    }
    if (raw_dump) { DUMP(c, "scores", layer, 0, raw_dump, H, S, T); free(raw_dump); }
    if (masked_dump) { DUMP(c, "scores_masked", layer, 0, masked_dump, H, S, T); free(masked_dump); }
    if (scores_dump) { DUMP(c, "softmax", layer, 0, scores_dump, H, S, T); free(scores_dump); }
    DUMP(c, "attn_pre_wo", layer, 0, att, S, qd);
    linear_core(att, layer_tensor(m, layer, "attention.wo.weight"), out, S, dim, qd, c->nthreads);   /* :522 */
    free(xq); free(xk); free(xv); free(att);
    return 0;
}

/* LlamaFeedForward.Forward, llamatransformer.go:593-624 */
static void ffn_forward(orc_ctx* c, int layer, const uint16_t* x, int S, uint16_t* out) {
    orc_model* m = c->m; int dim = m->a.dim, F = m->ffn_hidden;
    const float* silu = orc_silu_table();
    uint16_t* g = (uint16_t*)malloc((size_t)S * F * 2); uint16_t* u = (uint16_t*)malloc((size_t)S * F * 2);
    linear_core(x, layer_tensor(m, layer, "feed_forward.w1.weight"), g, S, F, dim, c->nthreads);
    linear_core(x, layer_tensor(m, layer, "feed_forward.w3.weight"), u, S, F, dim, c->nthreads);
    for (int64_t i = 0; i < (int64_t)S * F; i++) {
        uint16_t gs = trunc16(silu[g[i]]);                                     /* activations.go:36-39 */
        g[i] = trunc16(wide(gs) * wide(u[i]));                                 /* MultiplyElementwise :614 */
    }
    DUMP(c, "ffn_h", layer, 0, g, S, F);
    linear_core(g, layer_tensor(m, layer, "feed_forward.w2.weight"), out, S, dim, F, c->nthreads);
    free(g); free(u);
}

int orc_forward(orc_ctx* c, const int32_t* tokens, int S, int start_pos, float* logits_out, int32_t* argmax_last) {
    orc_model* m = c->m; const orc_args* a = &m->a; int dim = a->dim, V = a->vocab_size;
    if (!m->finalized) return fail("model not finalized");
    if (S == 0) return fail("empty token array");                               /* llamatransformer.go:146-148 */
    int T = start_pos + S;
    if (start_pos < 0 || T > m->cis_rows) return fail("incompatible locStart, locEnd values and tensor");   /* tensor.go:275-279 via :123 */
    if (T > c->seq_len) return fail("incompatible locStart, locEnd values and tensor");                   /* KV Slice :409 */
    if (S > 1 && T % S != 0) return fail("two tensor shapes cannot be broadcasted: [%d %d %d] and [%d %d]", a->n_heads, S, T, S, S);
    const uint16_t* emb = find_tensor(m, "tok_embeddings.weight")->data;
    uint16_t* x = (uint16_t*)malloc((size_t)S * dim * 2); uint16_t* n = (uint16_t*)malloc((size_t)S * dim * 2);
    uint16_t* t = (uint16_t*)malloc((size_t)S * dim * 2); uint16_t* h = (uint16_t*)malloc((size_t)S * dim * 2);
    for (int i = 0; i < S; i++) {                                                /* Fwd_Get_Rows impl:160-171 */
        if (tokens[i] < 0 || tokens[i] >= V) { free(x); free(n); free(t); free(h); return fail("token id %d out of range", tokens[i]); }
        memcpy(x + (size_t)i * dim, emb + (size_t)tokens[i] * dim, (size_t)dim * 2);
    }
    DUMP(c, "embedding", -1, 0, x, S, dim);
    for (int l = 0; l < a->n_layers; l++) {                                      /* LlamaTransformerBlock.Forward :215-254 */
        {   /* doNormalization's own output (trunc(x * rsqrt), llamatransformer.go:641-660) is a stage the reference's test pins too */
            uint16_t* pre = c->dump ? (uint16_t*)malloc((size_t)S * dim * 2) : NULL;
            orc_rmsnorm_bf16(x, layer_tensor(m, l, "attention_norm.weight"), n, S, dim, a->norm_eps, pre);
            if (pre) { DUMP(c, "attn_norm_part", l, 0, pre, S, dim); free(pre); }
        }
        DUMP(c, "attn_norm", l, 0, n, S, dim);
        attention_forward(c, l, n, S, start_pos, t);
        DUMP(c, "attn_out", l, 0, t, S, dim);
        for (int64_t i = 0; i < (int64_t)S * dim; i++) h[i] = trunc16(wide(x[i]) + wide(t[i]));   /* ml.Add :232 */
        DUMP(c, "h", l, 0, h, S, dim);
        orc_rmsnorm_bf16(h, layer_tensor(m, l, "ffn_norm.weight"), n, S, dim, a->norm_eps, NULL);
        ffn_forward(c, l, n, S, t);
        for (int64_t i = 0; i < (int64_t)S * dim; i++) x[i] = trunc16(wide(h[i]) + wide(t[i]));   /* ml.Add :248 */
        DUMP(c, "block_out", l, 0, x, S, dim);
    }
    orc_rmsnorm_bf16(x, find_tensor(m, "norm.weight")->data, n, S, dim, a->norm_eps, NULL);        /* :166 */
    DUMP(c, "final_norm", -1, 0, n, S, dim);
    const uint16_t* wout = find_tensor(m, "output.weight")->data;
    if (logits_out) {                                                            /* :170-175 all S rows */
        uint16_t* lg = (uint16_t*)malloc((size_t)S * V * 2);
        linear_core(n, wout, lg, S, V, dim, c->nthreads);
        for (int64_t i = 0; i < (int64_t)S * V; i++) logits_out[i] = wide(lg[i]);
        if (argmax_last) *argmax_last = orc_argmax_f32(logits_out + (size_t)(S - 1) * V, V);
        free(lg);
    } else {
        uint16_t* lg = (uint16_t*)malloc((size_t)V * 2); float* lf = (float*)malloc((size_t)V * 4);
        linear_core(n + (size_t)(S - 1) * dim, wout, lg, 1, V, dim, c->nthreads);
        for (int i = 0; i < V; i++) lf[i] = wide(lg[i]);
        if (argmax_last) *argmax_last = orc_argmax_f32(lf, V);
        free(lg); free(lf);
    }
    free(x); free(n); free(t); free(h);
    return 0;
}

/* inference.go:173-254 (greedy; stop ids handled by the caller) */
int orc_generate(orc_ctx* c, const int32_t* prompt, int prompt_len, int32_t* out_tokens, int n_out, double* secs_per_step) {
    if (prompt_len >= c->seq_len) return fail("context SequenceLength %d must be higher than prompt tokens length %d", c->seq_len, prompt_len);
    int32_t* tokens = (int32_t*)malloc((size_t)c->seq_len * 4);
    for (int i = 0; i < c->seq_len; i++) tokens[i] = -1;
    memcpy(tokens, prompt, (size_t)prompt_len * 4);
    int prev = 0, n = 0;
    for (int cur = prompt_len; cur < c->seq_len && n < n_out; cur++) {
        struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
        int32_t next;
        if (orc_forward(c, tokens + prev, cur - prev, prev, NULL, &next) != 0) { free(tokens); return -1; }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (secs_per_step) secs_per_step[n] = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
        if (tokens[cur] != -1) next = tokens[cur];
        tokens[cur] = next; out_tokens[n++] = next; prev = cur;
    }
    free(tokens);
    return n;
}
