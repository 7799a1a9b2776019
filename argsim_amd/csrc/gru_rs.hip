// gru_rs.hip -- BPTT of the cuDNN-style GRU recurrence (reference src/model.py:15,120-121,160) for D = 512 in the
// "reduce-scatter" decomposition.
//
// The LDS-weight team kernel of gru.hip (gru_bwd_team_kernel) gives a workgroup 16 hidden units j and lets it compute
//     dH_{p}[rows, its 16 j] = sum over ALL 1536 gate columns c' of dgh_{p+1}[rows, c'] R'[c'][j],
// so every workgroup reads every gate-gradient column of its rows each step: 16 x 1536 floats = 98 KB per team and step through
// the CU's vector-memory path -- three times the forward's exchange volume, and every load sits between the arrival of the
// last producer's data and the end of the step's MFMAs.
//
// Here a workgroup multiplies its OWN 48 gate-gradient columns -- which its gate threads have just computed, so the A operand
// goes through LDS and never leaves the CU -- by the 48 x 512 slice of R' it holds in LDS (the very slice the forward holds):
//     partial_{ht}[rows, all 512 j] = dgh_p[rows, its 48 c'] R'[its 48 c'][:]           (K = 48, N = 512)
// and the 32 partial rows are summed on the consumer's side: workgroup ht' reads the 16 columns it owns out of all 32
// producers' partials.  Per team and step 32 KB are stored and 32 KB loaded (8 x 16-byte loads per lane instead of 24), the
// K split over a team's waves and the LDS reduction of its partial sums are gone (the waves split N), and a team meets at
// ONE barrier per step instead of two.
//
// What it costs (measured): every row and step now WRITES 32 partial rows of 512 floats -- 64 KB of unique
// bytes against the 6 KB of dgh the other form stores once and lets 32 workgroups read from L2.  On a FULL 256 x 64 batch that is
// 8.2 GB of stores per training step: the kernel becomes store-bandwidth bound (3.7 ms against 3.0 ms; with the stores ablated
// 2.3 ms); small FULL batches lose as well (+2..5 %).  Where a workgroup's teams run out of rows at different times -- a ragged batch
// with a long tail: most steps belong to one or two lone chains -- the volume is small and the shorter dependency chain wins (batch
// 100 x 512 ragged: 40.0 ms per training step against 43.6).  The host picks the form per call from the fill it expects (model.cpp
// rs_pick: below 0.40); both forms compute the same sums up to fp32 summation order.
//
// Exchange ("the data is the flag", as in gru.hip, with a tag bit instead of a sentinel value so that the buffer can be a
// ring): a partial tile has exactly one consumer, so the exchange is a ring of TWO slots per 16-row block; the k-th use of a
// slot carries k & 1 in the least significant mantissa bit of every float (the launcher fills the ring with 1-bits; the
// first use expects 0).  A consumer re-loads until every dword it reads carries the expected bit.  Forcing that bit moves a
// partial sum by at most one unit in the last place (relative 6e-8; the gradients' tolerance is 2e-4) and is a pure function of
// the value, so the kernel stays deterministic.  Why two slots suffice: a producer overwrites slot s (production i + 2) only
// after it has consumed production i + 1 of EVERY workgroup of its chain group, and each of those was stored after its
// producer had consumed -- loaded and verified -- production i out of slot s.
// Layout of one job's ring: X[slot 2][16-row block B/16][consumer 32][row group 4][producer 32][column 16][row in group 4]
// floats: what one consumer wave needs (its 4 rows of the consumer's 16 columns from all 32 producers) is 8 KB of contiguous
// memory, one 1 KB load instruction per 4 producers; a producer wave stores one MFMA result tile (16 rows x 16 columns of
// one consumer) as it stands in its registers with one 16-byte store per lane.
#include <algorithm>
#include "kernels.h"
#include "gru_dev.h"

namespace avae {

constexpr int kRsBlock = 8192;        // floats per (slot, 16-row block, consumer): [g 4][producer 32][c 16][e 4]

__device__ __forceinline__ unsigned tag_bit(unsigned v, unsigned tag) { return (v & ~1u) | tag; }

template <bool PIPE, int T, bool CMP>
__global__ __launch_bounds__(1024, 4) void gru_bwd_rs_kernel(GruArgs a)
{
    constexpr int D = 512, HT = 32, KS = 16 / T, NT = 32 / KS, RB = 16 * T, AST = 52;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Wl = lds;                                        // [t 32][q 3][lane 64][4]: B fragments of R'[ht*48 + 16q + 4kh + e][16t + n]   96 KB
    float* Aimg = Wl + 96 * 256;                            // [team T][buf 2][row 16][AST]: this step's dgh of the team's rows, the A operand
    float* red = Aimg + T * 2 * 16 * AST;                   // [4][16] bias-gradient sums
    unsigned* sync = reinterpret_cast<unsigned*>(red + 64); // [team T] barrier counters, [team T] exchange-ready epochs

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int team, wk;                                           // wave -> (team, wave of the team); waves w and w + 4 share a SIMD
    if (T == 4) { team = wave >> 2; wk = wave & 3; }        // a team's waves on all four SIMDs
    else        { team = wave >> 3; wk = wave & 7; }
    const int n = lane & 15, kh = lane >> 4;
    const TeamMap tm = team_map(a, RB);
    const int cid = tm.cid, ht = tm.ht;
    const GruJob& J = a.job[tm.jb];
    const int Bs = a.B;                                     // rows per position of the external arrays
    const int B = (CMP && a.Bx > Bs) ? a.Bx : Bs;           // rows of the launch geometry (slots)
    xcd_publish(a.counters + 64 + cid, a.counters + 128 + cid * HT, ht);

    // weights -> LDS.  Read as whole rows of R' (coalesced), scattered into B-fragment order: element (c', j) of the slice
    // belongs to tile t = j / 16, lane (n = j % 16, kh = (c' / 4) % 4), block q = c' / 16, element e = c' % 4.
    for (int i = tid; i < 48 * (D / 4); i += 1024) {
        const int c = i / (D / 4), j4 = (i - c * (D / 4)) * 4;
        const float4 v = *reinterpret_cast<const float4*>(J.R + (size_t)(ht * 48 + c) * D + j4);
        const int q = c >> 4, kq = (c >> 2) & 3, e = c & 3, t = j4 >> 4, n0 = j4 & 15;
        float* dst = Wl + ((size_t)(t * 3 + q) * 64 + kq * 16 + n0) * 4 + e;
        dst[0] = v.x; dst[4] = v.y; dst[8] = v.z; dst[12] = v.w;
    }
    if (tid < 2 * T) sync[tid] = 0u;
    if (tid < 64) red[tid] = 0.f;
    const int tt = wk * 64 + lane, gn = tt & 15, gr = (tt >> 4) & 15;
    const bool gate_thread = tt < 256;                      // the team's first 256 threads own one (row, unit) each
    const bool gate_wave = wk < 4;
    const int j = ht * 16 + gn;
    const bool fast = __builtin_amdgcn_readfirstlane((int)group_same_xcd(a.counters + 64 + cid, a.counters + 128 + cid * HT, ht, HT, a.err, a.force_slow, true)) != 0;   // has a __syncthreads
    float* const timg = Aimg + team * (2 * 16 * AST);
    unsigned* tsync = sync + team;
    unsigned* ready = sync + T + team;
    unsigned epoch = 0;
    {
        const int tq = __builtin_amdgcn_readfirstlane(team);      // one arbitration order on all four SIMDs (see gru.hip)
        if (tq == 0) __builtin_amdgcn_s_setprio(3);
        else if (tq == 1) __builtin_amdgcn_s_setprio(2);
        else if (tq == 2) __builtin_amdgcn_s_setprio(1);
    }
    const float* __restrict__ p_sv = J.sv;
    const float* __restrict__ p_hp = J.hp;
    const float* __restrict__ p_do = J.dh_out;
    float* __restrict__ p_dgi = J.dgi;
    float* __restrict__ p_dghw = J.dgh;
    float* p_dh0 = J.dh0;
    const int* p_lens = a.slens ? a.slens : a.lens;
    const int* const perm = a.perm;
    int* p_err = a.err;
    const int j_rev = J.reverse, ldg = a.ldg, ldh = a.ldh;
    float* const xg = a.xbuf + (size_t)tm.jb * (size_t)B * (4 * kRsBlock);          // this job's ring: 2 slots x B/16 blocks x 32 consumers
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(xg);
    const __amdgpu_buffer_rsrc_t rs_sv = make_rsrc(p_sv), rs_hp = make_rsrc(p_hp), rs_do = make_rsrc(p_do ? p_do : p_hp);
    const __amdgpu_buffer_rsrc_t rs_dgi = make_rsrc(p_dgi), rs_dghw = make_rsrc(p_dghw);
    float sb_r = 0.f, sb_u = 0.f, sb_n = 0.f, sb_nr = 0.f;
    const bool want_dh0 = (p_dh0 != nullptr);
    const int p_last = want_dh0 ? -1 : 0;                     // p == -1: only dh0 = carry + (sum of the partials of step 0)
    unsigned it = 0;
    float carry_reg = 0.f;                                    // !PIPE: dH_{p+1} u_{p+1} of this thread's (row, unit)
    float carry_r0 = 0.f, carry_r1 = 0.f, carry_r2 = 0.f, carry_r3 = 0.f;      // PIPE: the same, per row block
    u32x4 hv[8];                                              // the consumer's 8 loads: producer 4i + (lane >> 4), rows 4 wk + e, column n
    // steps of this team's row blocks (padding skipped, gru_dev.h team_steps): non-increasing over r
    int nst0 = a.p_end, nst1 = a.p_end, nst2 = a.p_end, nst3 = a.p_end;
    if (a.slens) {
        nst3 = tm.nrb > 3 ? min(a.p_end, team_steps(a.slens, (tm.slot + 3 * tm.cpj) * RB + team * 16, lane)) : 0;
        nst2 = tm.nrb > 2 ? max(nst3, min(a.p_end, team_steps(a.slens, (tm.slot + 2 * tm.cpj) * RB + team * 16, lane))) : 0;
        nst1 = tm.nrb > 1 ? max(nst2, min(a.p_end, team_steps(a.slens, (tm.slot + 1 * tm.cpj) * RB + team * 16, lane))) : 0;
        nst0 = max(nst1, min(a.p_end, team_steps(a.slens, tm.slot * RB + team * 16, lane)));
    }
    auto nst_of = [&](int r) -> int { return r == 0 ? nst0 : (r == 1 ? nst1 : (r == 2 ? nst2 : nst3)); };
    // byte offset of this lane's first load of production `ic` of the 16-row block at row0 (consumer side)
    auto cons_off = [&](int ic, int row0) -> unsigned {
        return ((((unsigned)(ic & 1) * (unsigned)(B >> 4) + (unsigned)(row0 >> 4)) * 32u + (unsigned)ht) * (unsigned)kRsBlock + (unsigned)wk * 2048u + (unsigned)lane * 4u) * 4u;
    };
    auto issue8 = [&](unsigned off) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) hv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 1024 * i, 16);     // sc1: served by L2, never by this CU's L1
    };
    // every dword of the eight pieces carries the tag?  (uniform branch: one AND- or one OR-reduction, one instruction per dword)
    auto all_tagged = [&](unsigned tag) __attribute__((always_inline)) -> bool {
        unsigned m;
        if (tag) {
            m = 1u;
#pragma unroll
            for (int i = 0; i < 8; ++i) m &= (hv[i].x & hv[i].y) & (hv[i].z & hv[i].w);
            return !__any((m & 1u) == 0u);
        }
        m = 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i) m |= (hv[i].x | hv[i].y) | (hv[i].z | hv[i].w);
        return !__any((m & 1u) != 0u);
    };
    int len_g = j_rev ? p_lens[tm.slot * RB + team * 16 + gr] : 0;
    int ge_cur = tm.slot * RB + team * 16 + gr;                 // batch row of this thread's gate row in the current item
    if (perm) ge_cur = perm[ge_cur];
    const int* const rowmap = CMP ? a.rowmap : nullptr;
    const bool dgi_by_pos = rowmap != nullptr && J.dgi_by_pos != 0;
    if (a.slens && gate_thread && (!rowmap || dgi_by_pos)) {
        // positions behind a row block's steps: zero gate gradients (the weight-gradient GEMMs sum over every row)
        for (int r = 0; r < tm.nrb; ++r) {
            const int gs = (tm.slot + r * tm.cpj) * RB + team * 16 + gr, ge = perm ? perm[gs] : gs;
            if (CMP && ge >= Bs) continue;
            for (int p = nst_of(r); p < a.p_end; ++p) {
                const unsigned orow = (((unsigned)p * (unsigned)Bs + (unsigned)ge) * (unsigned)ldg + ht * 48 + gn * 3) * 4u;
                if (!rowmap) bstore3(0.f, 0.f, 0.f, rs_dghw, orow);
                bstore3(0.f, 0.f, 0.f, rs_dgi, orow);
            }
        }
    }
    const int p_first = nst0 - 1;
    for (int p = p_first; p >= p_last; --p) {
      for (int r = 0; r < tm.nrb && (p < 0 || p < nst_of(r)); ++r, ++it) {
        const int row0 = (tm.slot + r * tm.cpj) * RB + team * 16;
        const int nstr = nst_of(r);
        const bool have_next = p + 1 < nstr;                      // a step after this one exists: its partials are this step's dH
        const bool produce = p > p_last;                          // this step's partial is consumed at step p - 1
        const int ic = nstr - 2 - p, ip = nstr - 1 - p;           // production index consumed / produced here
        // PIPE: the item after this one
        const bool same_p = r + 1 < tm.nrb && (p < 0 || p < nst_of(r + 1));
        const int r2 = same_p ? r + 1 : 0, p2 = same_p ? p : p - 1;
        int len2g = len_g, ge2 = ge_cur;
        if constexpr (PIPE) {
            if (p2 >= p_last) {
                const int row2 = (tm.slot + r2 * tm.cpj) * RB + team * 16;
                if (j_rev) len2g = p_lens[row2 + gr];
                ge2 = perm ? perm[row2 + gr] : row2 + gr;
            }
        }
        // (1) exchange-independent loads of the gate phase
        const int gpos = pos_map(p < 0 ? 0 : p, len_g, j_rev);
        const unsigned prix = (unsigned)gpos * (unsigned)Bs + (unsigned)ge_cur;
        const bool inb = !CMP || ge_cur < Bs;
        const int crow = (rowmap && gate_thread && p >= 0) ? (inb ? rowmap[prix] : -1) : (int)prix;
        const bool real = !CMP || crow >= 0;
        const unsigned rix = (unsigned)crow;
        float4 sv = make_float4(0.f, 0.f, 0.f, 0.f); float s_hp = 0.f, s_do = 0.f;
        if (p >= 0 && gate_thread && real) {
            sv = bload4(rs_sv, ((rix * HT + ht) * 64 + gn * 4) * 4u);
            s_hp = bload1(rs_hp, (rix * D + j) * 4u);
            s_do = p_do ? bload1(rs_do, (rix * (unsigned)ldh + j) * 4u) : 0.f;
        }
        float s_carry = 0.f;
        if constexpr (PIPE) s_carry = r == 0 ? carry_r0 : (r == 1 ? carry_r1 : (r == 2 ? carry_r2 : carry_r3));
        else s_carry = carry_reg;
        // (2) dH of the step after this one, this workgroup's 16 columns: the sum of the 32 producers' partials
        float carried = 0.f;
        if (have_next && gate_wave) {
            const unsigned tag = (unsigned)(ic >> 1) & 1u;
            const unsigned off = cons_off(ic, row0);
            SpinGuard sg;
            // start signal (heuristic): lane l reads the last dword producer l & 31 stores for these 16 rows; the check of every
            // loaded dword below is what correctness rests on
            const unsigned poff = off - ((unsigned)wk * 2048u + (unsigned)lane * 4u) * 4u + (3u * 2048u + (unsigned)(lane & 31) * 64u + 63u) * 4u;
            auto probe = [&]() __attribute__((always_inline)) {
                for (;;) {
                    unsigned v = tag;
                    if (lane < 32) v = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_x, (int)poff, 0, 16);
                    if (!__any((v & 1u) != tag) || sg.expired(p_err)) break;
                }
            };
            if constexpr (!PIPE) {
                // ONE load site: the first polling pass polls with one wave per team (a poll costs ~180 texture-addresser cycles), which
                // releases its partners through LDS; a pass that still met an old tag is repeated by the wave itself
                // (GruArgs::spec -- few rows alive, the step is one chain's latency: pass 0 loads at once, without the probe's round
                //  trip in front; a pass that met an old tag goes on to the polling passes)
                for (int att = a.spec ? 0 : 1;; ++att) {
                    if (att >= 1 && (wk == 0 || att > 1)) probe();
                    if (att == 1) {
                        if (wk == 0) { if (lane == 0) __hip_atomic_store(ready, it + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                        else while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < it + 1u) __builtin_amdgcn_s_sleep(1);
                    }
                    issue8(off);
                    if (all_tagged(tag) || sg.expired(p_err)) break;
                }
                if (wk == 0 && lane == 0) __hip_atomic_store(ready, it + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (a pass 0 that succeeded: partners that did not are waiting)
            } else {
                while (!all_tagged(tag) && !sg.expired(p_err)) { probe(); issue8(off); }
            }
            // fixed summation order: the eight pieces of a lane, then the four lane groups
            float s[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                auto val = [&](int i) -> float { return __uint_as_float(hv[i][e] & ~1u); };      // (the tag bit cleared: an exact zero stays one)
                const float x0 = val(0) + val(1), x1 = val(2) + val(3), x2 = val(4) + val(5), x3 = val(6) + val(7);
                s[e] = (x0 + x1) + (x2 + x3);
            }
            // lanes l, l ^ 16, l ^ 32, l ^ 48 hold the same 4 elements (rows 4 wk + e of column n) of different producers: reduce-scatter
            // them so that lane (kh, n) ends with element e = kh -- the (row, unit) its gate thread owns
            const bool hi = (lane & 32) != 0, odd = (lane & 16) != 0;
            float k0 = hi ? s[2] : s[0], k1 = hi ? s[3] : s[1];
            const float t0 = hi ? s[0] : s[2], t1 = hi ? s[1] : s[3];
            k0 += __shfl_xor(t0, 32, 64); k1 += __shfl_xor(t1, 32, 64);
            const float keep = odd ? k1 : k0, send = odd ? k0 : k1;
            carried = keep + __shfl_xor(send, 16, 64);
        }
        // PIPE: the next item is another chain whose partials every producer stored an item ago: its loads go in flight now,
        // behind the gate math, the barrier and the MFMAs of this item (verified at its own item)
        if constexpr (PIPE) {
            asm volatile("" :: "v"(sv.x), "v"(sv.y), "v"(sv.z), "v"(sv.w), "v"(s_hp), "v"(s_do), "v"(len2g), "v"(ge2));
            __builtin_amdgcn_sched_barrier(0);
            if (p2 >= p_last && gate_wave && p2 + 1 < nst_of(r2))
                issue8(cons_off(nst_of(r2) - 2 - p2, (tm.slot + r2 * tm.cpj) * RB + team * 16));
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // (3) gate derivatives: the team's 256 threads, one (row, unit) each
        const int buf = (int)(it & 1u);
        if (gate_thread) {
            carried += s_carry;
            if (p < 0) { if (inb) p_dh0[(size_t)ge_cur * D + j] = carried; }
            else {
                const float dH = carried + s_do;
                const float r_ = sv.x, u = sv.y, nn = sv.z;
                const float dn = dH * (1.f - u) * (1.f - nn * nn);
                const float du = dH * (s_hp - nn) * u * (1.f - u);
                const float dr = dn * sv.w * r_ * (1.f - r_);
                const float dnr = dn * r_;
                if (produce) {      // the A operand of this step's product: position c' = 3 gn + gate of row gr
                    float* ap = timg + (buf * 16 + gr) * AST + gn * 3;
                    ap[0] = dr; ap[1] = du; ap[2] = dnr;
                }
                const unsigned orow = (rix * (unsigned)ldg + ht * 48 + gn * 3) * 4u;
                const unsigned orow_i = dgi_by_pos ? (prix * (unsigned)ldg + ht * 48 + gn * 3) * 4u : orow;
                if (real) bstore3(dr, du, dnr, rs_dghw, orow);                 // the row-major copy the weight-gradient GEMM reads
                if (real || (dgi_by_pos && inb)) bstore3(dr, du, dn, rs_dgi, orow_i);
                const float c1 = dH * u;
                if constexpr (PIPE) { if (r == 0) carry_r0 = c1; else if (r == 1) carry_r1 = c1; else if (r == 2) carry_r2 = c1; else carry_r3 = c1; }
                else carry_reg = c1;
                sb_r += dr; sb_u += du; sb_n += dn; sb_nr += dnr;
            }
        }
        // (4) partial_{ht}[16 rows, 512] = dgh[16 rows, 48] R'[48, 512]: this wave's NT tiles of 16 columns, K = 48 in 12 steps
        __builtin_amdgcn_sched_barrier(0);
        if (produce) {
            epoch += KS; team_barrier(tsync, epoch);                // the team's A image is complete (the only barrier of the step)
            const unsigned tagp = (unsigned)(ip >> 1) & 1u;
            const unsigned sbase = ((((unsigned)(ip & 1) * (unsigned)(B >> 4) + (unsigned)(row0 >> 4)) * 32u) * (unsigned)kRsBlock
                                    + (unsigned)kh * 2048u + (unsigned)ht * 64u + (unsigned)n * 4u) * 4u;
            f32x4 af[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) af[q] = *reinterpret_cast<const f32x4*>(timg + (buf * 16 + n) * AST + 16 * q + 4 * kh);
#pragma unroll
            for (int hf = 0; hf < NT / 4; ++hf) {
                const int t0 = wk * NT + hf * 4;
                f32x4 acc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    f32x4 b[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const f32x4*>(Wl + ((size_t)((t0 + i) * 3 + q) * 64 + lane) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q][e], b[i][e], acc[i], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x4 v = {tag_bit(__float_as_uint(acc[i][0]), tagp), tag_bit(__float_as_uint(acc[i][1]), tagp),
                                     tag_bit(__float_as_uint(acc[i][2]), tagp), tag_bit(__float_as_uint(acc[i][3]), tagp)};
                    const unsigned so = sbase + (unsigned)(t0 + i) * (unsigned)(kRsBlock * 4);
                    if (fast) __builtin_amdgcn_raw_buffer_store_b128(v, rs_x, (int)so, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(v, rs_x, (int)so, 0, 16);
                }
            }
        }
        len_g = len2g; ge_cur = ge2;
      }
    }
    // bias gradients: all rows of the workgroup into LDS, then one atomic per (gate, unit)
    if (J.dbW || J.dbR) {
        if (gate_thread) { atomicAdd(&red[gn], sb_r); atomicAdd(&red[16 + gn], sb_u); atomicAdd(&red[32 + gn], sb_n); atomicAdd(&red[48 + gn], sb_nr); }
        __syncthreads();
        if (tid < 48) {
            const int gate = tid >> 4, u = tid & 15;
            if (J.dbW) atomicAdd(J.dbW + ht * 48 + u * 3 + gate, red[gate * 16 + u]);
            if (J.dbR) atomicAdd(J.dbR + ht * 48 + u * 3 + gate, red[(gate == 2 ? 3 : gate) * 16 + u]);
        }
    }
}

size_t gru_bwd_rs_xbuf_floats(int njobs, int rows) { return (size_t)njobs * (size_t)rows * (4 * kRsBlock); }

// fill value of the ring: every tag bit set (the first use of a slot expects 0)
hipError_t gru_bwd_rs_launch(hipStream_t st, const GruArgs& a, int T, int C, bool pipe, bool cmp)
{
    const int lds_bytes = (96 * 256 + T * 2 * 16 * 52 + 64) * 4 + 64;
    if (T == 4) {
        if (pipe) return cmp ? launch_team(st, gru_bwd_rs_kernel<true, 4, true>, a, lds_bytes, C) : launch_team(st, gru_bwd_rs_kernel<true, 4, false>, a, lds_bytes, C);
        return cmp ? launch_team(st, gru_bwd_rs_kernel<false, 4, true>, a, lds_bytes, C) : launch_team(st, gru_bwd_rs_kernel<false, 4, false>, a, lds_bytes, C);
    }
    if (pipe) return cmp ? launch_team(st, gru_bwd_rs_kernel<true, 2, true>, a, lds_bytes, C) : launch_team(st, gru_bwd_rs_kernel<true, 2, false>, a, lds_bytes, C);
    return cmp ? launch_team(st, gru_bwd_rs_kernel<false, 2, true>, a, lds_bytes, C) : launch_team(st, gru_bwd_rs_kernel<false, 2, false>, a, lds_bytes, C);
}

}  // namespace avae
