// kernels.h -- internal launch interface of the gfx950 kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace avae {

// ---------------------------------------------------------------- GEMM (gemm_f32.hip)
// C[M,N] = alpha * op(A) * op(B) (+ bias[n]) (+ C if accumulate), exact fp32 on
// v_mfma_f32_32x32x2_f32.  Row-major storage with leading dimensions:
//   a_mc == false : A[m*lda + k]   (k contiguous)     a_mc == true : A[k*lda + m]
//   b_nc == false : B[n*ldb + k]   (k contiguous)     b_nc == true : B[k*ldb + n]
// The contiguous dimension and every leading dimension must be a multiple of 4 floats and
// every base pointer 16-byte aligned.  M/N/K edges are predicated (zero filled).
// dyn (optional, device): the real row count when it is only known on the device:
//   dyn_kind 1 -> M_eff = min(M, *dyn) (row tiles beyond it exit),  2 -> K_eff = min(K, *dyn).
// split_k > 1: grid.z slices of K, combined with float atomics INTO C (C must hold the value to
// accumulate onto, e.g. zeros); bias is added by slice 0.
struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc;
    float alpha;
    int accumulate;      // C += ...
    int split_k;
    const int* dyn; int dyn_kind;
    int dyn_expect;      // dyn_kind 1: the count the host expects (an earlier call's; 0 = unknown) -- launch shaping only, never correctness
    int thin;            // 1: 32x128 block tiles (thin row panels) instead of 128x128; 2: 64x64; 3: skinny form (32x32 per workgroup, K split over its waves)
    // optional second problem of identical shape, layout and scalars, run by the same launch (grid.y = 2): two
    // weight-gradient GEMMs over the same rows then share ONE round of workgroups at half the K split (half the
    // float-atomic traffic each), two small affines share one launch.  Exact-fp32 kernel only.
    const float* A2; const float* B2; float* C2; const float* bias2;
    float* slab = nullptr; size_t slab_floats = 0;      // bf16 TN form, K split: workspace for the slices' partial tiles (summed by a reduce pass instead of float atomics)
    unsigned short* c16 = nullptr;      // bf16 NT form, phased kernel only (gemm_bf16_c16_ok): the result goes HERE as an fp16 panel [M][ldc] INSTEAD of fp32 C (the logits of a
                                        // training step: softmax_ce reads the 2-byte panel and turns it into the bf16 gradient in place); no bias, no accumulate, no K split
    int nt8 = 1;         // bf16 NT form: the phased LDS-DMA kernel (gemm_bf16_p8.hip) where the shape allows; 0: gemm_bf16_nt256_kernel
};
hipError_t gemm_f32(hipStream_t st, bool a_mc, bool b_nc, const GemmArgs& g);
// same contract on the bf16 matrix cores: every fp32 operand element is split into three bf16 in registers and
// six partial products are accumulated in fp32 (gemm_f32s.hip; fp32-accurate, 128x128 tiles only: g.thin ignored)
hipError_t gemm_f32s(hipStream_t st, bool a_mc, bool b_nc, const GemmArgs& g);
// bf16-operand GEMM from fp32 operands in any layout (gemm_f32s.hip, one plane): RNE to bf16 while staging into LDS, one
// product, fp32 accumulate; no conversion pass (g.thin ignored)
hipError_t gemm_bf16_direct(hipStream_t st, bool a_mc, bool b_nc, const GemmArgs& g);

// ---------------------------------------------------------------- bf16-operand GEMM (gemm_bf16.hip)
// fp32 -> bf16 panel: transpose == false: src [rows][cols] (ld) -> dst [rows][ldd];  transpose == true:
// src [K][X] (ld) -> dst [X][ldd] with ldd >= K.  ldd is a multiple of 8; pad columns are zero filled.
hipError_t cvt_bf16(hipStream_t st, const float* src, int ld, bool transpose, int rows_or_K, int cols_or_X,
                    unsigned short* dst, int ldd);
// C = alpha * A B^T (+bias)(+C): A [M][lda], B [N][ldb] bf16 k-contiguous; the other fields as in GemmArgs
hipError_t gemm_bf16_nt(hipStream_t st, const unsigned short* A, int lda, const unsigned short* B, int ldb, const GemmArgs& g);
// the phased 256x256 form of the same contract (gemm_bf16_p8.hip); _ok: K % 64 == 0, N % 4 == 0, aligned C / bias, no device-side K
bool gemm_bf16_p8_ok(const GemmArgs& g, int lda, int ldb);
hipError_t gemm_bf16_p8(hipStream_t st, const unsigned short* A, int lda, const unsigned short* B, int ldb, const GemmArgs& g, int s2);
// would gemm_bf16_nt run this problem on the phased kernel (the only one that writes GemmArgs::c16)?
bool gemm_bf16_c16_ok(const GemmArgs& g, int lda, int ldb);
// ... and of gemm_bf16_tn's contract (operands [k][x] row-major, K host- or device-side, any K)
bool gemm_bf16_p8_tn_ok(const GemmArgs& g, int lda, int ldb);
hipError_t gemm_bf16_p8_tn(hipStream_t st, const unsigned short* A, int lda, const unsigned short* B, int ldb, const GemmArgs& g, int s2);
// C = alpha * A^T B (+C): A [K][lda] (m contiguous), B [K][ldb] (n contiguous) bf16, read with transposing LDS loads -- no
// transposed copy of either operand (M, N, lda, ldb multiples of 8; split_k > 1: float atomics into C)
hipError_t gemm_bf16_tn(hipStream_t st, const unsigned short* A, int lda, const unsigned short* B, int ldb, const GemmArgs& g);
// bf16 [K][X] (row stride ld) -> bf16 [X][ldk] (ldk = K rounded up to 8, pad zero-filled)
hipError_t transpose_bf16(hipStream_t st, const unsigned short* src, int ld, int K, int X, unsigned short* dst, int ldk);

// ---------------------------------------------------------------- GRU (gru.hip)
// Gate-interleaved layout ("G16"): the 3D gate rows of W, R, bW, bR and the 3D columns of
// gi / dgi / dgh are stored in the order  c' = ht*48 + u*3 + gate   (ht = unit/16, u = unit%16,
// gate: 0=r 1=u 2=n) so that the 48 values one workgroup owns are contiguous and the three gates of
// one unit are adjacent (one 12-byte access per thread in the gate phase).
constexpr int kMaxGruJobs = 3;
struct GruJob {
    const float* gi;      // + job column offset; row (pos*B + b) at gi + (pos*B+b)*ldg
    const int32_t* gi_rows;   // optional (team kernels only): row gi_rows[pos*B + b] instead -- a layer fed by embedding rows reads
                          // its projection straight out of the per-id table (model.cpp use_table), no per-token copy
    const float* R;       // (3D, D) G16 rows
    const float* bR;      // (3D) G16
    const float* h0;      // (B, D) or nullptr (zeros)
    float* hs;            // output, row (pos*B+b) at hs + (pos*B+b)*ldh (+ column offset applied)
    float* sv;            // saved r,u,n,hn: (S,B,HT,16,4) or nullptr
    float* hp;            // saved h_prev (S,B,D) or nullptr
    int reverse;          // 1: time index map of tf.reverse_sequence (needs lens)
    // optional, bf16 team kernels only: the row-major h (indexed like hs, ldh) / h_prev (S,B,D) copies as bf16 INSTEAD of fp32
    unsigned short* hs16;
    unsigned short* hp16; //   (the backward reads h_prev from here when given)
    // backward only
    const float* dh_out;  // grad wrt hs (same indexing as hs: ldh) or nullptr
    float* dgi;           // (S,B,ldg) G16 columns (+ job offset)
    float* dgh;           // (S,B,ldg)
    unsigned short* dgi16; // optional, bf16 team kernels only: dgi / dgh as bf16 (S,B,ldg) row-major INSTEAD of the fp32 arrays
    unsigned short* dgh16; //   (gru_backward_uses_team tells whether the launch will honour them)
    float* dh0;           // (B,D) or nullptr
    float* carry;         // (B,D) scratch: dH_{p+1} * u_{p+1}
    int dgi_by_pos;       // with GruArgs::rowmap: dgi / dgi16 keep the padded (pos * B + row) layout, zeros at padding (a table-fed layer: summed by token id)
    float* dbW;           // (3D) G16 += column sums of dgi (or nullptr)
    float* dbR;           // (3D) G16 += column sums of dgh (or nullptr)
};
struct GruArgs {
    GruJob job[kMaxGruJobs];
    int njobs, S, B, D, ldg, ldh;
    const int* lens;      // (B) or nullptr
    // optional, team kernels only (gru.hip "Padding skipped"): steps each SLOT really has and the batch row sitting in it;
    // both or neither.  The other kernel forms ignore them and run every step of every row.
    const int* slens;     // (B) by slot
    const int* perm;      // (B) slot -> batch row
    // optional, team kernels only: COMPACT external arrays.  rowmap[pos * B + row] = the row of gi / hs / sv / hp / dh_out / dgi / dgh
    // that position has among the real (non-padding) positions in time-major order, -1 for a padding position (nothing is read
    // or written for it; it still takes part in the exchange).  gi_rows (a table-fed layer) stays indexed by pos * B + row.
    const int* rowmap;
    // optional, team kernels only, with slens + perm + rowmap: the launch geometry's batch, a multiple of the 16 T-row block above B.
    // Slots hold Bx rows (slens / perm have Bx entries, the exchange scratch and the carry Bx rows); a slot whose perm entry is >= B is
    // a PHANTOM row (slens 1): nothing external is read or written for it.  Every external array keeps B rows per position.  0: B.
    int Bx;
    int G;                // batch groups per job
    int rows_per_group;   // multiple of 16
    int p_begin, p_end;   // steps [p_begin, p_end) of this launch
    unsigned* counters;   // 1664 words (step counters | detection counters | XCD ids), zeroed by the launcher
    int* err;             // device error word (set on spin timeout)
    unsigned long long* stamps;   // 32 words of diagnostic phase sums (ablate bit 32) or nullptr
    int item_pipeline;    // 1: use the software-pipelined D = 512 forward kernel where its geometry applies
    float* xbuf;          // scratch for the team kernels' exchange in MFMA-tile order (see gru.hip "tiled exchange"): at least
    size_t xbuf_floats;   // njobs*S*B*D floats (forward) / njobs*S*B*3D (backward); nullptr / too small: register-form kernels
    int stagger;          // 1: delay the second half of the grid by ~half a step (co-resident chains de-phased)
    int force_slow;       // 1: never use the same-XCD L2 fast path
    int sv16;             // 1 (bf16 team kernels, forward AND backward of a layer): the saved gates r, u, n, hn are stored as bf16
    int bf16;             // 1 (compute_dtype 1): the team kernels round both operands of the recurrent product to bf16
    int spec;             // 1 (team kernels, one row block per workgroup): few rows are alive per step -- a consumer loads its operand at once, without a probe round trip in front (speed only)
    int bwd_rs;           // 1: the backward runs the reduce-scatter team kernel (gru_rs.hip) where the team geometry applies and xbuf holds its ring
    int ablate;           // timing experiments only: 1 no MFMA/A loads, 2 no gate-phase loads, 4 no saves, 8 cheap activations, 16 no sync
};
hipError_t gru_forward(hipStream_t st, const GruArgs& a, bool persistent);
bool gru_forward_uses_team(const GruArgs& a, bool persistent);      // true: the launch runs the LDS-weight team kernels (gi_rows honoured)
hipError_t gru_backward(hipStream_t st, const GruArgs& a, bool persistent);
// geometry of the team kernels a launch of this shape would run: T teams of 16 rows per 16 T-row block, cpj chain groups
// (workgroups) per job and hidden tile, nrb row blocks per workgroup; false: another kernel form runs (no slens / perm)
bool gru_team_shape(const GruArgs& a, bool fwd, bool persistent, int* T, int* cpj, int* nrb);
int gru_team_batch(int B);        // smallest row count >= B (within 256) the team kernels have a geometry for with one job and with two; 0: none
bool gru_backward_uses_team(const GruArgs& a, bool persistent);     // true: the launch runs the LDS-weight team kernels (dgi16 / dgh16 honoured in bf16 mode)
bool gru_dim_supported(int D);
// reduce-scatter form of the backward team kernels (gru_rs.hip): floats of exchange scratch it needs for njobs jobs over `rows` slots
size_t gru_bwd_rs_xbuf_floats(int njobs, int rows);
hipError_t gru_bwd_rs_launch(hipStream_t st, const GruArgs& a, int T, int C, bool pipe, bool cmp);
bool gru_backward_uses_rs(const GruArgs& a, bool persistent);     // true: the launch runs the reduce-scatter form
// one GRU step from a zero state for B rows (the top encoder layer's backward direction: gru.hip "one step from a zero
// state"): gi (B, 3D) / bR G16; h -> h_out[b * ldo + j]; sv (B, D, 4) = r, u, n, hn or nullptr
hipError_t gru_first_step_fwd(hipStream_t st, const float* gi, const float* bR, float* h_out, int ldo, float* sv, int B, int D, const int32_t* lens = nullptr);      // lens: rows of length 0 get h = 0
// its backward: dh[b * ldd + j] -> dgi, dgh (B, 3D) G16
hipError_t gru_first_step_bwd(hipStream_t st, const float* dh, int ldd, const float* sv, float* dgi, float* dgh, int B, int D, const int32_t* lens = nullptr);
bool gru_diag_build();     // true: built with -DAVAE_DIAG (ablation / stamp instantiations present)

// ---------------------------------------------------------------- small kernels (ops.hip)
struct PrepArgs {
    const int32_t* src; const int32_t* tgt;   // (B,Ss),(B,St) row-major
    int B, Ss, St, eos, bos;
    int train; float keepwd; uint64_t seed; const uint8_t* keep_mask;  // (St,B)
    int32_t* src_tm;    // (Ss,B)
    int32_t* lens_src;  // (B)
    int32_t* lens_tgt;  // (B)
    int32_t* lead;      // (St+1,B)
    int32_t* gold;      // (St+1,B)
    int32_t* rank;      // (St+1,B) compact row of (t,b) or -1
    int32_t* cidx;      // (N) flat (t*B+b) of compact row
    int32_t* ntok;      // [0]=N
    float* zero2;       // optional: two floats cleared here (the loss accumulators), saves a memset launch
    int32_t* chunk_counts;   // scratch, kPrepChunks ints: kept positions per chunk of the flat time-major order
};
constexpr int kPrepChunks = 1024;
hipError_t prep_ids(hipStream_t st, const PrepArgs& p);

hipError_t embed_gather(hipStream_t st, const float* E, const int32_t* ids, float* out, int n, int D, int V);
// dE[ids0[t]] += dout0[t], dE[ids1[t]] += dout1[t]; scratch: embed_scatter_scratch_ints(n0 + n1, V) ints
hipError_t embed_scatter_add2(hipStream_t st, float* dE, const int32_t* ids0, const float* dout0, int n0, const int32_t* ids1,
                              const float* dout1, int n1, int D, int V, int32_t* scratch);
// token groups of an id source (ops.hip "token groups"): built once per step, used by the forward gather and the backward sums
bool id_groups_supported(int V);
inline size_t id_groups_ints(size_t n, size_t V) { return 4 * V + 2 + n + (n / 32 + V + 1) + 2 * V + 4; }
hipError_t id_groups_build(hipStream_t st, const int32_t* ids, int n, int V, int32_t* scratch, bool lists);
const int32_t* id_groups_rank(int32_t* scratch, int n, int V);     // [V]  index among the present ids, -1 absent
const int32_t* id_groups_uid(int32_t* scratch, int n, int V);      // [V]  present ids, ascending
const int32_t* id_groups_count(int32_t* scratch, int n, int V);    // [1]  how many
// out[t] = rank[ids[t]]
hipError_t rank_rows(hipStream_t st, int32_t* out, const int32_t* ids, const int32_t* rank, int n, int V);
hipError_t rows_gather_ranked(hipStream_t st, float* dst, const float* src, const int32_t* ids, const int32_t* rank, int n, int W, int V);
hipError_t rows_group_sum(hipStream_t st, float* dst, const int32_t* ids, const float* src, int n, int W, int V, int32_t* scratch);
hipError_t rows_add_indexed(hipStream_t st, float* dst, const float* src, const int32_t* uid, const int32_t* nuniq, int n_max, int D);
inline size_t embed_scatter_scratch_ints(size_t n, size_t V) { return 4 * V + 2 + n + (n / 32 + V + 1); }
// Row order for the padding-skipping team kernels: rows sorted by steps[b] = lens[b] + add (descending, stable) and dealt
// in groups of 16 over the workgroups of a (T, cpj) team geometry so that every workgroup's row blocks get shorter with r
// (slot of sorted group g: ((g % cpj) + (g / cpj / T) * cpj) * T + (g / cpj) % T).  perm[slot] = batch row, slens[slot] =
// its steps.  Up to three geometries in one launch.
struct RowOrder { const int32_t* lens; int add, T, cpj; int32_t* perm; int32_t* slens; };
hipError_t row_order(hipStream_t st, const RowOrder* orders, int n, int Breal, int B, int S, int32_t* steps_sum = nullptr, int sum_rows = 0);      // steps_sum[0..1] <- sum of the first order's steps, sum_rows; B >= Breal slots: the rows beyond Breal are phantom rows of one step
// dst[i,:] = src[idx[i],:] for i < *n_dev
hipError_t rows_gather(hipStream_t st, float* dst, const float* src, const int32_t* idx, const int32_t* n_dev, int n_max, int D, const int32_t* map = nullptr);
// dst[r,:] = rank[r] >= 0 ? src[rank[r],:] : 0   for r < rows
hipError_t rows_expand(hipStream_t st, float* dst, const float* src, const int32_t* rank, int rows, int D, const int32_t* map = nullptr);
// h[b,:] = hs[(len_b-1)*B + b, :]  (model.py:135)
hipError_t pick_last(hipStream_t st, float* h, const float* hs, const int32_t* lens, int B, int W, const int32_t* map = nullptr);      // (map: compact rows, row_map)
// the same from a bf16 source: h[b,:] = float(hs16[(len_b-1)*B + b, :])
hipError_t pick_last16(hipStream_t st, float* h, const unsigned short* hs16, const int32_t* lens, int B, int W, const int32_t* map = nullptr);
// dhs[(len_b-1)*B+b,:] += d[b,:]
hipError_t pick_last_add(hipStream_t st, float* dhs, const float* d, const int32_t* lens, int B, int W, const int32_t* map = nullptr);
// compact row map of an id source (ops.hip "compact row map"): map (S*B), nact (S scratch), count [1]
hipError_t row_map(hipStream_t st, const int32_t* lens, int add, int S, int B, int32_t* map, int32_t* nact, int32_t* count);
hipError_t zero_rows_dyn(hipStream_t st, float* X, const int32_t* count, int rows_max, int W);      // X[r,:] = 0 for r < min(rows_max, *count)
// dhs = 0 everywhere except dhs[(len_b-1)*B+b,:] = dh[b,:]
hipError_t pick_last_bwd(hipStream_t st, float* dhs, const float* dh, const int32_t* lens, int S, int B, int W);

// z = mu (+ exp(lv/2)*eps when train; eps from eps_in or the counter RNG, echoed to eps_out);
// kld[i] = 0.5(mu^2 + e^lv - lv - 1); acc[0] += sum max(kld, free_bits)
hipError_t latent_fwd(hipStream_t st, const float* mu, const float* lv, const float* eps_in, float* eps_out, float* z,
                      float* kld, int n, int train, uint64_t seed, float free_bits, float* acc);
// dmu = dz + c*mu ; dlv = dz*eps*0.5*exp(lv/2) + c*0.5*(exp(lv)-1), c = anneal/(Bg*R) (free-bits gated)
hipError_t latent_bwd(hipStream_t st, const float* dz, const float* mu, const float* lv, const float* eps_used,
                      float* dmu, float* dlv, int B, int R, float coef, float free_bits);

struct CeArgs {
    float* logits;            // (N,V) in: logits, out (if dlogits): (softmax - onehot) * scale
    const int32_t* gold;      // (T,B) flat
    const int32_t* cidx;      // (N) -> flat (t,b)
    const int32_t* n_dev;     // real N
    int n_max, V;
    int write_grad; float inv_n;   // scale; <= 0 -> 1 / *n_dev
    float* loss_samp; float* errt_samp; int32_t* pred;   // (N) optional
    float* loss_acc;          // [0] += sum loss_samp
    unsigned short* grad16;   // optional (N,V) bf16: with write_grad the gradient goes HERE (the bf16 GEMM's operand panel) and the logits stay
    int logits16;             // the logits are the fp16 panel the GEMM left IN grad16 (GemmArgs::c16): read from there, overwritten in place by the bf16 gradient
};
hipError_t softmax_ce(hipStream_t st, const CeArgs& a);
// pred[i] = argmax_j logits[i, j]  (first max), rows < n
hipError_t argmax_rows(hipStream_t st, const float* logits, int32_t* pred, int n, int V);

// out[n] (+)= sum_m X[m, n]
hipError_t colsum(hipStream_t st, const float* X, int M, int N, int ldx, float* out, const int32_t* m_dev);
hipError_t add3(hipStream_t st, float* out, const float* a, const float* b, const float* c, int64_t n);
hipError_t zero_fill(hipStream_t st, void* p, size_t bytes);      // a plain kernel instead of the runtime's blit (16-byte aligned p, bytes % 4 == 0)

struct AdamArgs { float* p; const float* g; float* m; float* v; int64_t n; float lr_t, b1, b2, eps;
                  const int* skip_if; };   // device word: non-zero -> the update is a no-op (GRU time-out flag)
hipError_t adam_tf(hipStream_t st, const AdamArgs& a);

// ---------------------------------------------------------------- greedy decoding (decode.hip)
// the whole loop of model.py:204-219 in one persistent launch; every pointer is device memory
struct DecodeArgs {
    const float* E;                                 // (V, D) tied embedding
    const float *W[8], *R[8], *bW[8], *bR[8];       // decoder GRU layers, G16 row order
    const float *Kout, *bout;                       // out affine: kernel (D, D) as (in, out), bias (D)
    float* state[2];                                // ping-pong (L, b, D); [0] holds state_in on entry
    float* o;                                       // (b, D)
    float* part_val; int32_t* part_idx;             // (G, b) partial argmax per workgroup, G = CU count
    int32_t* ids_tm;                                // (steps + 1, b) time-major ids; row 0 = bos on entry
    int32_t* out_ids;                               // (b, steps) result, eos beyond the kept tokens
    int32_t* kept;                                  // [0] <- tokens kept per row
    unsigned* bar;                                  // grid-barrier counter, zero on entry
    int* err;                                       // error word (spin time-out)
    int b, steps, D, V, L, eos, cache_e; float isd;
};
// returns hipErrorInvalidValue where the geometry does not fit (D / CUs > 2 units per workgroup, LDS), the caller
// then falls back to one launch sequence per token; *grid_out = workgroups launched
hipError_t decode_greedy(hipStream_t st, DecodeArgs a, int* grid_out);
int decode_workgroups();                            // CU count of the current device (size of part_val / part_idx rows)

// natural <-> G16 row permutation of a (3D, cols) matrix (cols = 1 for biases)
hipError_t g16_permute(hipStream_t st, float* dst, const float* src, int D, int cols, bool to_g16);

// losses[0..2] = loss_gen, loss_kld, loss (model.py:181-185) from the per-token CE (n = min(n_max, *n_dev) entries) and the
// per-element KL terms (nk entries, each floored at free_bits), summed in a fixed order: bit-reproducible
hipError_t finalize_losses(hipStream_t st, float* losses, const float* loss_samp, const int32_t* n_dev, int n_max,
                           const float* kld, int nk, float free_bits, float inv_br, float anneal);

}  // namespace avae
