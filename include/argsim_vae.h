/* argsim_vae.h -- C ABI of libargsim_vae.so, the MI355X (gfx950) implementation of the
 * argsim/argsim sequence-VAE hot path.
 *
 * The reference has no FFI: its de-facto boundary is the Record returned by vAe()
 * (reference src/model.py:73,191) and the feed->fetch pairs its callers use.  Every entry
 * point below names the reference call site it replaces (paths relative to the reference
 * repository root).  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - opaque handle; one handle = one GPU = one host thread.
 *   - every tensor pointer is a DEVICE pointer owned by the caller (e.g. a PyTorch-ROCm
 *     tensor's data_ptr()); the library never frees caller memory.  It owns only its
 *     workspace.  Parameters, gradients and Adam slots live in four caller-allocated flat
 *     float32 buffers bound with avae_bind_state().
 *   - int return: 0 = ok, non-zero = error, text via avae_last_error().  No C++ exception
 *     crosses the ABI.
 *   - all work is enqueued on the stream given to avae_set_stream() (default: the null
 *     stream) and is asynchronous unless the entry returns host scalars.
 *   - ids are row-major (B, S) int32, eos-padded, exactly as util_np.vpack makes them
 *     (src/util_np.py:5-13).  z is row-major (b, dim_rep) float32 as np.save'd by
 *     src/eval_embed_reason.py:41.
 */
#ifndef ARGSIM_VAE_H
#define ARGSIM_VAE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct avae_ctx* avae_handle;

/* model section of src/config.json:10-21 + vAe() keyword defaults (src/model.py:48-66) */
typedef struct avae_config {
    int32_t dim_tgt;      /* vocab size V                      */
    int32_t dim_emb;      /* model dim D (16,32,64,128,256,512) */
    int32_t dim_rep;      /* latent dim R (multiple of 4)      */
    int32_t rnn_layers;   /* L                                 */
    float   accelerate;   /* schedule speed, model.py:77       */
    float   learn_rate;   /* model.py:80                       */
    int32_t bos, eos;     /* model.py:65-66                    */
    int32_t max_batch;    /* workspace sizing hints; grown on  */
    int32_t max_len;      /*   demand if exceeded              */
    /* extensions (not in the reference; identity at beta=1, free_bits=0) */
    float   kl_beta;      /* multiplies rate_anneal            */
    float   free_bits;    /* per-dimension KL floor (nats)     */
    int32_t compute_dtype;/* 0 = exact fp32 (reference);       */
                          /* 1 = bf16 operands in every contraction (GEMMs, GRU recurrence), fp32 accumulate/state/weights (BASELINE configs[2]) */
                          /* 2 = fp32 GEMMs on the bf16 matrix cores: each operand element split into 3 bf16 */
                          /*     in registers, 6 partial products, fp32 accumulate (fp32-accurate)          */
} avae_config;

/* kind selector for avae_get_tensor / avae_set_tensor */
enum { AVAE_PARAM = 0, AVAE_GRAD = 1, AVAE_ADAM_M = 2, AVAE_ADAM_V = 3 };

/* ---- lifetime -------------------------------------------------------------------------- */
/* replaces graph construction vAe(mode, **C)  (src/train.py:74,85; src/eval_embed_reason.py:24) */
int  avae_create(const avae_config* cfg, int device, avae_handle* out);
void avae_destroy(avae_handle h);
const char* avae_last_error(avae_handle h);           /* h may be NULL: last create error */
int  avae_set_stream(avae_handle h, void* hip_stream);

/* vae.mu.shape[1] etc. (src/explore_infer.py:33) */
int  avae_get_dims(avae_handle h, int32_t* V, int32_t* D, int32_t* R, int32_t* L);

/* ---- state (tf.train.Saver over global variables: src/train.py:92-96,121) --------------- */
int64_t avae_state_numel(avae_handle h);              /* floats in each flat buffer          */
int  avae_bind_state(avae_handle h, float* params, float* grads, float* adam_m, float* adam_v);
int  avae_param_count(avae_handle h);
const char* avae_param_name(avae_handle h, int i);
/* natural (checkpoint) shape of a named variable; internal storage may be permuted */
int  avae_param_info(avae_handle h, const char* name, int64_t* offset, int32_t* ndim, int64_t shape[4]);
/* copy a named variable between its natural layout (device buffer `buf`) and the flat state */
int  avae_get_tensor(avae_handle h, const char* name, int kind, float* buf);
int  avae_set_tensor(avae_handle h, const char* name, int kind, const float* buf);
/* global_step (src/model.py:76, src/train.py:119) */
int  avae_get_step(avae_handle h, int64_t* step);
int  avae_set_step(avae_handle h, int64_t step);
/* rate_keepwd, rate_anneal, rate_update at the current step (src/model.py:78-80) */
int  avae_get_schedule(avae_handle h, float out[3]);

/* ---- training: sess.run(model_train.train_step)  (src/train.py:118, src/model.py:189) --- */
/* src, tgt     (B,S_src),(B,S_tgt) int32 device
 * seed         counter-RNG key for word-dropout and the reparameterisation draw
 * keep_mask    optional (S_tgt,B) uint8 device, time-major: overrides random_uniform<keepwd
 * eps          optional (B,R) float device: overrides random_normal
 * n_tok_global tokens N of the GLOBAL batch (data-parallel exactness, mean over N in
 *              model.py:181); <= 0 means "this call's own N"
 * b_global     rows of the global batch; <= 0 means B                                  */
int  avae_forward_backward(avae_handle h, const int32_t* src, const int32_t* tgt,
                           int32_t B, int32_t S_src, int32_t S_tgt, uint64_t seed,
                           const uint8_t* keep_mask, const float* eps,
                           float n_tok_global, float b_global);
/* TF-style Adam (epsilon outside the bias-corrected sqrt) + global_step += 1 */
int  avae_adam_step(avae_handle h);
/* = avae_forward_backward + avae_adam_step */
int  avae_train_step(avae_handle h, const int32_t* src, const int32_t* tgt,
                     int32_t B, int32_t S_src, int32_t S_tgt, uint64_t seed,
                     const uint8_t* keep_mask, const float* eps);
/* loss_gen, loss_kld, loss of the last forward (synchronises the stream) */
int  avae_get_losses(avae_handle h, float out[3]);
/* data-parallel hook, called on the host while backward is being ENQUEUED.
 *   bucket >= 0: grads[offset, offset+count) is final (no later kernel of this step writes it).  The call is made
 *                right after the next persistent GRU launch has been enqueued (or at the end of backward), so a
 *                collective the callee orders behind the compute stream's current tail (event + side stream) runs
 *                beside the GEMM phase that follows that launch.
 *   bucket == AVAE_HOOK_FENCE (offset = count = 0): a persistent GRU launch comes next; the callee must make the
 *                compute stream wait for every collective it has in flight (the launch needs all CUs resident and
 *                must not share the device with a kernel that may wait on a peer GPU).                          */
enum { AVAE_HOOK_FENCE = -1 };
typedef void (*avae_grad_hook)(void* user, int bucket, int64_t offset, int64_t count);
int  avae_set_grad_hook(avae_handle h, avae_grad_hook hook, void* user);
/* The buckets the hook speaks of: contiguous ranges of the flat gradient buffer, indexed in buffer order
 *   0 decode/out | 1..L decode/rnn/lL..l1 | L+1 latent | L+2..2L+1 encode/rnnL..rnn1 | 2L+2 embed/embedding
 * (src/model.py:108-168 variable scopes).  ANNOUNCEMENT order within a step is fixed and does not depend on
 * the batch shape -- ranks whose shards differ in width pair their collectives by call order:
 *   0, 1, .., 2L (encode/rnn2), 2L+2 (embedding), 2L+1 (encode/rnn1)
 * i.e. buffer order except that the embedding bucket, complete once the first encoder layer's input gradient is
 * scattered, goes out before that layer's weight-gradient GEMMs and overlaps them.                               */
int  avae_bucket_count(avae_handle h);
int  avae_bucket_info(avae_handle h, int i, int64_t* offset, int64_t* count);

/* ---- validation: (errt_samp, loss_gen_samp, loss_kld_samp)  (src/train.py:109-110) ------ */
/* mode 'valid': no word dropout, z = mu.  outputs: errt_samp,loss_gen_samp (>= B*(S_tgt+1))
 * float device; loss_kld_samp (B,R) float device; *n_out = N (host; synchronises)         */
int  avae_eval(avae_handle h, const int32_t* src, const int32_t* tgt,
               int32_t B, int32_t S_src, int32_t S_tgt,
               float* errt_samp, float* loss_gen_samp, float* loss_kld_samp, int32_t* n_out);

/* ---- inference ------------------------------------------------------------------------- */
/* model.z.eval({model.src: data})  (src/model.py:194-201; src/eval_embed_reason.py:38,50):
 * z = mu (b,R); lv_out optional                                                          */
int  avae_encode(avae_handle h, const int32_t* src, int32_t b, int32_t t, float* z_out, float* lv_out);
/* vae.state_in.eval({vae.z: z})  (src/model.py:213): (L,b,D)                              */
int  avae_decode_init(avae_handle h, const float* z, int32_t b, float* state_out);
/* sess.run((vae.pred, vae.state_ex), {vae.lead: x, vae.state_in: s})  (src/model.py:216)  */
int  avae_decode_step(avae_handle h, const int32_t* lead, const float* state_in, int32_t b,
                      int32_t* pred_out, float* state_out);
/* the whole greedy loop of decode() (src/model.py:204-219) without a host round trip per
 * token: out_ids (b, steps) int32 device, *n_steps = tokens kept per row (host)           */
int  avae_decode_greedy(avae_handle h, const float* z, int32_t b, int32_t steps,
                        int32_t* out_ids, int32_t* n_steps);

#ifdef __cplusplus
}
#endif
#endif /* ARGSIM_VAE_H */
