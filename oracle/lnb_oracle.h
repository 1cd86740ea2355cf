/*
 * lnb_oracle.h -- CPU ORACLE for the LlamaTransformer.Forward hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the arithmetic of
 * adalkiran/llama-nuts-and-bolts (pure Go, CPU) for the path
 *   src/model/llamatransformer.go:145-180  (LlamaTransformer.Forward)
 * and everything under it in src/ml + src/dtype.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it; the product library
 * (liblnb_hip.so) never links, imports or calls anything in oracle/.
 *
 * PARITY PINNING: the Go reference cannot be built here (no Go toolchain) and the
 * Llama-3.1-8B checkpoint is absent, so this oracle is pinned against the
 * weight-free known-answer tests the reference ships (src/dtype/bfloat16_test.go,
 * src/ml/operations_test.go, src/ml/tensor_test.go) and the RoPE frequency table
 * printed in docs/10-ROPE-ROTARY-POSITIONAL-EMBEDDINGS.md:273-288,397-407
 * (tests/test_oracle_kat.py).  The real-weight goldens of
 * src/model/llamatransformer_simulated_test.go cannot be replayed: end-to-end
 * parity is "pinned on KATs, unpinned on real-weight token ids".
 *
 * Every function cites the reference file:line it restates.
 */
#ifndef LNB_ORACLE_H
#define LNB_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- src/dtype/bfloat16.go ------------------------------------------------ */
uint16_t orc_f32_to_bf16(float f);          /* :31-33,:59-61  bits>>16 (truncate) */
float    orc_bf16_to_f32(uint16_t b);       /* :19-21,:55-57  bits<<16            */

/* ---- synthetic model generator (spec in DESIGN.md "Synthetic weights") ---- */
/* kind 0: sigma*g (projection / embedding matrices), kind 1: 1 + 0.1*g (norms) */
uint16_t orc_synth_bf16(uint64_t seed, uint32_t tensor_id, uint64_t idx, int kind, float sigma);
int32_t  orc_synth_token(uint64_t seed, uint64_t i, int32_t vocab);
void     orc_synth_fill(uint16_t* dst, uint64_t n, uint64_t seed, uint32_t tensor_id, int kind, float sigma);

/* ---- src/ml ops (small generic forms, used by the KAT tests) -------------- */
/* operations_lineartransform.go:37-70,145-207 : y[m,n] = trunc(sum_k x[m,k]*W[n,k]) */
void orc_linear_bf16(const uint16_t* x, const uint16_t* w, uint16_t* y, int rows, int n_out, int k_in, int nthreads);
/* operations_lineartransform.go:72-103 (f32 variant, result stays f32 then ToBFloat16 is NOT applied: dst F32->ToBFloat16 at :205 applies to both;
   the reference returns bf16 for both variants; we return the pre-truncation f32 as well for the F32 KAT) */
void orc_linear_f32(const float* x, const float* w, float* y_f32, uint16_t* y_bf16, int rows, int n_out, int k_in);
/* operations_matmul.go:24-60,136-182 : C[b,m,n] = trunc(sum_k A[b,m,k]*B[b,k,n]) */
void orc_matmul_bf16(const uint16_t* a, const uint16_t* b, uint16_t* c, int batch, int m, int k, int n);
/* operations_impl.go:11-24 */
int  orc_arange_bf16(int start, int end, int step, uint16_t* out);
int  orc_arange_f32(int start, int end, int step, float* out);
/* operations_impl.go:26-53 */
void orc_outer_bf16(const uint16_t* v1, int n1, const uint16_t* v2, int n2, uint16_t* out);
/* operations_impl.go:100-140 : out[2*i]=re, out[2*i+1]=im (complex64) */
void orc_polar_f32(const float* abs_, const float* angle, float* out_c64, int n);
void orc_polar_bf16(const uint16_t* abs_, const uint16_t* angle, float* out_c64, int n);
/* operations_impl.go:175-195 (dst zero-initialised, copy where j-i >= diagonal) */
void orc_triu_bf16(const uint16_t* in, uint16_t* out, int rows, int cols, int diagonal);
/* operations_impl.go:197-217 (bf16 in -> f32 out) */
void orc_pow_bf16(const uint16_t* in, float* out, int n, double power);
/* operations_impl.go:219-253 (f32, last dim) */
void orc_mean_f32(const float* in, float* out, int groups, int last);
/* operations_impl.go:478-511 (f32 in/out, f64 exp and sum, no max subtraction) */
void orc_softmax_f32(const float* in, float* out, int rows, int cols);
void orc_set_exp_impl(int which);      /* 0: host libm exp (default); 1: Go's portable math.Exp restated (fdlibm e_exp) -- softmax and the SiLU table */
double orc_exp_f64(double x);           /* the exp in effect */
/* operations_impl.go:513-548 */
int32_t orc_argmax_f32(const float* in, int n);
/* activations.go:10-25 */
const float* orc_silu_table(void);
/* llamatransformer.go:633-660 : full RMSNorm on [rows,dim] bf16 */
void orc_rmsnorm_bf16(const uint16_t* x, const uint16_t* w, uint16_t* y, int rows, int dim, float eps, uint16_t* pre_weight_opt);
/* llamatransformer.go:662-751 : freqs (after optional scaling, bf16) and the cis table [rows][dim/2][2] f32 */
void orc_rope_freqs(int head_dim, double theta, int use_scaled, uint16_t* freqs_out);
void orc_rope_table(int head_dim, int rows, double theta, int use_scaled, float* cis_out, uint16_t* angles_bf16_opt);
/* llamatransformer.go:753-790 : x [S, n_heads, head_dim] bf16 in place; cis rows for positions start..start+S-1 */
void orc_rope_apply(uint16_t* x, int S, int n_heads, int head_dim, const float* cis_rows);

/* ---- model + forward -------------------------------------------------------- */
typedef struct {
    int dim, n_layers, n_heads, n_kv_heads, vocab_size, multiple_of;
    double ffn_dim_multiplier;     /* <= -1 : unset  (llamatransformer.go:573-575) */
    float norm_eps;
    int use_scaled_rope;
    double rope_theta;
    int max_seq_len;               /* modelargs.go:23,42 : 2048 ; table rows = 2*max_seq_len */
} orc_args;

typedef struct orc_model orc_model;
typedef struct orc_ctx orc_ctx;

int  orc_ffn_hidden_dim(const orc_args* a);                 /* llamatransformer.go:569-577 */
orc_model* orc_model_create(const orc_args* a);
void orc_model_destroy(orc_model* m);
/* names are the Meta checkpoint keys (llamatransformer.go:84,98,105,191,202,273-283,580-587);
   data is COPIED (reference layout [out,in] row-major bf16).  returns 0 / -1 */
int  orc_model_set_tensor(orc_model* m, const char* name, const uint16_t* data, int64_t nelem);
const uint16_t* orc_model_get_tensor(orc_model* m, const char* name, int64_t* nelem);
void orc_model_fill_synthetic(orc_model* m, uint64_t seed, int nthreads);
int  orc_model_finalize(orc_model* m);                      /* builds RoPE table (llamatransformer.go:109) */
const float* orc_model_rope_table(orc_model* m, int* rows);

orc_ctx* orc_ctx_create(orc_model* m, int seq_len);         /* inferencecontext.go:17-46 (zero-filled) */
void orc_ctx_destroy(orc_ctx* c);
void orc_ctx_set_threads(orc_ctx* c, int nthreads);
const uint16_t* orc_ctx_cache(orc_ctx* c, int layer, int which /*0=K,1=V*/);

/* stage dump hook: called with a stage name, layer (-1 outside blocks), dtype (0 bf16,1 f32), shape */
typedef void (*orc_dump_fn)(void* user, const char* stage, int layer, int dtype, const void* data, const int64_t* shape, int rank);
void orc_ctx_set_dump(orc_ctx* c, orc_dump_fn fn, void* user);

/* llamatransformer.go:145-180.  logits_out: [S, vocab] f32 (may be NULL -> only argmax of last row
   is produced).  Returns 0, or <0 with orc_last_error(). */
int  orc_forward(orc_ctx* c, const int32_t* tokens, int S, int start_pos, float* logits_out, int32_t* argmax_last);
/* inference.go:173-254 greedy loop (no stop ids): writes seq_len - prompt_len tokens at most n_out */
int  orc_generate(orc_ctx* c, const int32_t* prompt, int prompt_len, int32_t* out_tokens, int n_out, double* secs_per_step_opt);
const char* orc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
