// Fused backward of the feed-forward half of the post-norm encoder layer, 16-bit modes, E = 128 / F = 256 -- and, compiled
// a second time with -DMIVIT_WIDTH64 (elem.h), the reference's shipped E = 64 / F = 128: the four-wave kernel and attn_out_bwd
// are written against the derived constants below (the numbers in the comments are those of width 128)
// (reference helpers/models.py:72-77 FeedForward, :104-106 x = LN2(x1 + ff(x1)); autograd of that block):
//
//   in : dy = dL/dx2 (x2 = gamma2 * n2 + beta2), n2 = xhat of LN2, rstd2, n1 = xhat of LN1 (x1 = gamma1 * n1 + beta1)
//   out: dx1 = dL/dx1,  dW1, db1, dW2, db2, dgamma2, dbeta2
//
// reading dy, n2, n1 ONCE and writing dx1 ONCE: the hidden activations h, their gradient dh and dz2 (the gradient of
// the pre-norm sum) never reach HBM -- h is recomputed from n1.  One persistent workgroup per CU (4 waves, each owning a
// SIMD's whole register file) walks 32-row tiles:
//   phase 0 (all threads, element-wise): LayerNorm backward dy, n2 -> dz2 (LDS image DZ), n1 -> LDS image X; the
//            column sums for dgamma2 / dbeta2 / db2 stay in registers; next tile's rows are already in flight.
//   phase 1 (wave w owns hidden units 64w .. 64w+63; its slices of fc1 and fc2^T live in REGISTERS):
//            u = n1 W1'^T + b1' (W1' = W1 * gamma1, b1' = b1 + W1 beta1: the LayerNorm affine is folded),
//            dh = (dz2 W2) * act'(u) -- both un-transposed, so the accumulators (lane = hidden unit, registers = rows)
//            ARE the row-contraction operands of the weight gradients:  dW1 += dh^T n1,  dW2 += dz2^T h, whose other
//            operand is a transposing LDS read (ds_read_b64_tr_b16) of X / DZ.  128 + 128 accumulator registers per
//            lane hold the wave's [64 x 128] and [128 x 64] gradient blocks for the whole launch.  dh -> LDS image DH.
//   phase 2 (wave w owns input features 32w .. 32w+31): dx1^T = W1^T dh^T + dz2^T from an LDS image of W1^T.
// Per-workgroup partial gradients go to slabs; a deterministic slab reduction finishes them (no atomics).
#include "common.h"
#include "stream_prims.h"
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <stdio.h>
#include <type_traits>
#include "elem.h"          // element type of this translation unit (bf16, or f16 under -DMIVIT_ELEM_F16): after every other include

#define RC(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

namespace {

#ifdef MIVIT_WIDTH64
constexpr int E = 64, F = 128;
constexpr int LDE = E + 16, LDF = F + 16;            // 160-byte rows: 10 x 16 B (b128 reads: a group of 16 lanes = 8 rows x 2 chunks, conflict-free
                                                     // for pitches of 2, 6, 10, 14 mod 16 units) and 40 banks (transposing reads: 8 rows, 8 banks apart)
#else
constexpr int E = 128, F = 256;
constexpr int LDE = E + 16, LDF = F + 16;            // LDS row pitches (elements): conflict-free b128 and transposing reads
#endif
constexpr int NW = 4, NT = NW * 64, R = 32;          // waves per workgroup, threads, rows per tile
constexpr int KS = E / 32, ET = E / 16;              // contraction steps over / 16-feature tiles of an embedding row
constexpr int CPR = E / 8;                           // 16-byte chunks per row = threads per row of the element-wise phase (16 | 8)
constexpr int RPP = NT / CPR, RPT = R / RPP;         // rows per pass of all threads (16 | 32), rows per thread (2 | 1)
constexpr int FW = F / NW, NTW = FW / 16;            // hidden units per wave (64 | 32) and their 16-unit tiles (4 | 2)
constexpr int EW = E / NW, KT2 = EW / 16;            // features per wave of the data-gradient phases (32 | 16), their tiles (2 | 1)
constexpr int FS = F / 32;                           // contraction steps over the hidden units
constexpr int NROWST = 2 * KT2;                      // 8-byte row stores per wave and full tile: what the tile wait counts on

constexpr int OFF_W1 = 0;                            // [F][LDE]  W1 as stored ([hidden unit][input feature]): serves u (plain reads) and dx1 (transposing reads)
constexpr int OFF_X = OFF_W1 + F * LDE * 2;          // [R][LDE]  x1 = gamma1 * n1 + beta1 rows
constexpr int OFF_DZ = OFF_X + R * LDE * 2;          // [R][LDE]  dz2 rows
constexpr int OFF_DH = OFF_DZ + R * LDE * 2;         // [F][R]    dh TRANSPOSED: one 64-byte row of the tile's 32 rows per hidden unit (see dht_off)
constexpr int OFF_VEC = OFF_DH + F * R * 2;          // b1 [F], gamma2 [E], gamma1 [E], beta1 [E] fp32
constexpr int OFF_ACC = OFF_VEC + (F + 3 * E) * 4;               // [6][NT] float4: running column sums of the element-wise phase (dgamma2, dbeta2, db2)
constexpr int OFF_STG = OFF_ACC + 6 * NT * 16;        // [3 RPT][NW][64] x 16 B: next tile's dy / n2 / n1 chunks, landed by LDS-DMA (thread-private slots)
constexpr int OFF_RST = OFF_STG + 3 * RPT * NT * 16; // [RPT][NT] floats: rstd2 of the next tile's rows per thread (LDS-DMA, thread-private slots)
constexpr int LDS_BYTES = OFF_RST + RPT * NT * 4;
// (Width 64 fits two workgroups per CU -- 75 KB of LDS, 190 registers x 4 waves.  Measured at the Framerate shape, 127 k rows:
//  grids of 512 give 67.2 against 65.2 us per launch, and twice the slab traffic: not used.)
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

// slab layout per workgroup (floats)
// (the order of the parameter arena: fc1.weight, fc1.bias, fc2.weight, fc2.bias, norm2.weight, norm2.bias -- when the
//  six outputs are contiguous like that, ONE reduction launch finishes all of them)
constexpr int SL_W1 = 0, SL_B1 = SL_W1 + F * E, SL_W2 = SL_B1 + F, SL_B2 = SL_W2 + E * F, SL_G2 = SL_B2 + E, SL_BE2 = SL_G2 + E,
              SL_TOTAL = SL_BE2 + E;

struct MlpBwdArgs {
    const bf16 *dy, *n2; const float *rstd2, *gamma2;
    const bf16 *n1; const float *gamma1, *beta1;
    const bf16 *W1; const float *b1; const bf16 *W2;
    int M;
    bf16 *dx1;
    float *slabs;          // [gridDim.x][SL_TOTAL]
    unsigned long long *dbg;   // -DMIVIT_PHASE_TIMING builds (scripts/phase_timing.py): per-wave cycle totals of the phases; else null
};
#ifdef MIVIT_PHASE_TIMING
#define PT_MARK(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pt[k] += t_ - pt_last; pt_last = t_; } while (0)
#else
#define PT_MARK(k) do { } while (0)
#endif

__device__ __forceinline__ bf16x8 pack8(const f32x4 a, const f32x4 b) {
    bf16x8 f;
    f[0] = (__bf16)a[0]; f[1] = (__bf16)a[1]; f[2] = (__bf16)a[2]; f[3] = (__bf16)a[3];
    f[4] = (__bf16)b[0]; f[5] = (__bf16)b[1]; f[6] = (__bf16)b[2]; f[7] = (__bf16)b[3];
    return f;
}
__device__ __forceinline__ float x4_sum(float v) { return rows4_sum(v); }   // common.h: permlane-swap / DPP forms
// sum over the CPR threads that share a row in the element-wise phase (a 16-lane row, or one half of it: xor 1, xor 2 by
// quad permutation, then the mirrored half-row)
__device__ __forceinline__ float g16_sum(float v) {
    if constexpr (CPR == 16) return row16_sum(v);
    v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); return v + dpp_mov<0x141>(v);
}
__device__ __forceinline__ bf16x8 lds_frag(const bf16 *p) { return *reinterpret_cast<const bf16x8 *>(p); }
__device__ __forceinline__ void unpack8(const uint4 &u, float (&v)[8]) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = elem_lo(w[i]); v[2 * i + 1] = elem_hi(w[i]); }
}

// dh image [hidden unit][tile row], unpadded 64-byte rows; the two 32-byte halves (tile rows 0-15 | 16-31) of hidden units 4..7
// (mod 8) are exchanged: a transposing read touches 8 consecutive image rows x 32 bytes per 32 lanes, which then cover 256
// distinct bytes.  The accumulators of dh = dz2 W2 hold 4 consecutive tile rows of ONE hidden unit per lane -> one 8-byte store
// (the [row][hidden] image of round 2 took four 2-byte stores per accumulator: 32 ds_write_b16 + ~190 VALU per tile and wave,
// 28 % of the kernel's LDS cycles in bank conflicts).  The kernel uses this map with the lane-dependent part factored out by
// hand (dst_lane / dbase below): written through this function the compiler re-derives one address register per (ks, rt)
// and hoists all 32 of them out of the tile loop -- 36 spilled registers.
__device__ __forceinline__ int dht_off(int hidden, int row) { return hidden * R + ((((row >> 4) ^ (hidden >> 2)) & 1) << 4) + (row & 15); }

// rows past the end were clamped to a real row: their staged chunks are zeroed IN REGISTERS.  (Written as `ok ? stg[i] : zero`
// the compiler selects between the LDS pointer and the address of a constant zero and loads through the result: a generic
// pointer, i.e. flat_load -- vmcnt AND lgkmcnt, out of order, so s_waitcnt vmcnt(0) in the middle of the element-wise phase.)
__device__ __forceinline__ uint4 keep_if(const uint4 v, bool ok) {
    const unsigned m = ok ? 0xffffffffu : 0u;
    return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
}
template <int ACT>
__global__ __launch_bounds__(NT) void mlp_block_bwd_kernel(const MlpBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16 *W1i = reinterpret_cast<bf16 *>(smem + OFF_W1);
    bf16 *X = reinterpret_cast<bf16 *>(smem + OFF_X), *DZ = reinterpret_cast<bf16 *>(smem + OFF_DZ), *DH = reinterpret_cast<bf16 *>(smem + OFF_DH);
    float *b1f = reinterpret_cast<float *>(smem + OFF_VEC);
    f32x4 *cacc = reinterpret_cast<f32x4 *>(smem + OFF_ACC);
    uint4 *stg = reinterpret_cast<uint4 *>(smem + OFF_STG);
    float *rst = reinterpret_cast<float *>(smem + OFF_RST);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, cq = lane & 15, q = cq >> 2, pp = cq & 3;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    // ---- one-time staging ----
    // (four loads in flight per thread: one load -> wait -> store per iteration is 16 serialised L2 round trips per workgroup)
    static_assert((F * (E / 8)) % (4 * NT) == 0, "W1 staging batches");
    for (int i0 = tid; i0 < F * (E / 8); i0 += 4 * NT) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * NT, n = i / (E / 8), cc = i - n * (E / 8);
            v[u] = *reinterpret_cast<const uint4 *>(a.W1 + (int64_t)n * E + cc * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * NT, n = i / (E / 8), cc = i - n * (E / 8);
            *reinterpret_cast<uint4 *>(W1i + n * LDE + cc * 8) = v[u];
        }
    }
    for (int n = tid; n < F; n += NT) b1f[n] = a.b1[n];
    for (int i = tid; i < E; i += NT) { b1f[F + i] = a.gamma2[i]; b1f[F + E + i] = a.gamma1[i]; b1f[F + 2 * E + i] = a.beta1[i]; }
    // register-resident operand of this wave's hidden units n = 64 wave + 16 nt + cq:
    //   w2f[nt][ks]: W2[32ks + 8g .. +7][n]           (column operand of dh = dz2 W2)
    bf16x8 w2f[NTW][KS];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int n = FW * wave + 16 * nt + cq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = __builtin_bit_cast(__bf16, a.W2[(int64_t)(ks * 32 + 8 * g + e) * F + n].v);
            w2f[nt][ks] = t;
        }
    }
    // this thread's slice of the element-wise phase: columns 8c .. 8c+7 of rows r0 (and r0 + 16) of the tile
    const int c = tid % CPR, r0 = tid / CPR;
    // LDS-typed pointers: as plain `const float *` they were kept across the tile loop as generic addresses and every read became
    // a flat_load (vmcnt AND lgkmcnt, out of order -> s_waitcnt vmcnt(0) in the element-wise phase, i.e. a wait for the previous
    // tile's store acks)
    typedef __attribute__((address_space(3))) const float lds_cf;
    lds_cf *gam2 = (lds_cf *)(b1f + F + 8 * c), *gam1 = (lds_cf *)(b1f + F + E + 8 * c), *bet1 = (lds_cf *)(b1f + F + 2 * E + 8 * c);
#pragma unroll
    for (int i = 0; i < 6; ++i) cacc[i * NT + tid] = f32x4{0.f, 0.f, 0.f, 0.f};       // (thread-private slots: no barrier needed)
    f32x4 dW1[NTW][ET], dW2[ET][NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int j = 0; j < ET; ++j) { dW1[i][j] = zero; dW2[j][i] = zero; }
    float db1[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) db1[i] = 0.f;
    __syncthreads();

    const int ntiles = (a.M + R - 1) / R;
    int tile = blockIdx.x;
    // next tile's rows travel global -> LDS by DMA into slots only this thread reads back: no registers held across the
    // compute phases, no barrier -- the issuing wave's vmcnt wait is the only ordering needed
    auto prefetch = [&](int t) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int row = min(t * R + r0 + RPP * i, a.M - 1);        // (64-bit only in the address)
            const int64_t o = (int64_t)row * E + 8 * c;
            dma16_opaque(a.dy + o, stg + (3 * i + 0) * NT + wave * 64);
            dma16_opaque(a.n2 + o, stg + (3 * i + 1) * NT + wave * 64);
            dma16_opaque(a.n1 + o, stg + (3 * i + 2) * NT + wave * 64);
            dma4_opaque(a.rstd2 + row, rst + i * NT + wave * 64);
        }
    };
    if (tile < ntiles) prefetch(tile);
    bool first = true;
#ifdef MIVIT_PHASE_TIMING
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_last = __builtin_amdgcn_s_memtime();
#endif
    for (; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * R;
        // ---------------- phase 0: LayerNorm backward (element-wise), images X and DZ ----------------
        {
            float sg[8], sb[8], sz[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) sg[e] = sb[e] = sz[e] = 0.f;
            // this thread's constants: six 16-byte LDS reads issued as ONE batch before the wait for the staged rows, the running
            // column sums as one batch of six at the end (read one float at a time where they are used -- and read-modify-written
            // one by one -- they were 28 serialised LDS round trips per tile)
            typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
            const f32x4 g2a = *(lds_cf4 *)(gam2), g2b = *(lds_cf4 *)(gam2 + 4), g1a = *(lds_cf4 *)(gam1), g1b = *(lds_cf4 *)(gam1 + 4);
            const f32x4 b1a = *(lds_cf4 *)(bet1), b1b = *(lds_cf4 *)(bet1 + 4);
            const float g2v[8] = {g2a[0], g2a[1], g2a[2], g2a[3], g2b[0], g2b[1], g2b[2], g2b[3]};
            const float g1v[8] = {g1a[0], g1a[1], g1a[2], g1a[3], g1b[0], g1b[1], g1b[2], g1b[3]};
            const float b1v[8] = {b1a[0], b1a[1], b1a[2], b1a[3], b1b[0], b1b[1], b1b[2], b1b[3]};
            // VM program order of a wave: P(t) [8 DMA pieces: the staged rows of tile t] | 4 row stores of tile t-1's phase 2 | this
            // wait.  P(t) was requested one tile ago; younger than it are only those four stores (every full tile issues them,
            // scripts/isa_check.py counts them in the ISA; the one partial tile is the last of the launch): vmcnt(4) = P(t) landed.
            // (width 64: 4 pieces, NROWST = 2 stores)
            if (first) wait_vm<0>(); else wait_vm<NROWST>();
            first = false;
            PT_MARK(0);                                            // 0: wait for the staged rows
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                float d[8], nh[8], gdy[8];
                const bool ok = row0 + r0 + RPP * i < a.M;          // rows past the end were clamped to a real row: cancel them
                const uint4 pdy = keep_if(stg[(3 * i + 0) * NT + tid], ok), pn2 = stg[(3 * i + 1) * NT + tid];
                const uint4 pn1 = keep_if(stg[(3 * i + 2) * NT + tid], ok);
                unpack8(pdy, d); unpack8(pn2, nh);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    gdy[e] = d[e] * g2v[e];
                    s1 += gdy[e]; s2 += gdy[e] * nh[e];
                    sg[e] += d[e] * nh[e]; sb[e] += d[e];
                }
                s1 = g16_sum(s1) * (1.f / E); s2 = g16_sum(s2) * (1.f / E);
                float dz[8];
                const float prs = __uint_as_float(__float_as_uint(rst[i * NT + tid]) & (ok ? 0xffffffffu : 0u));
#pragma unroll
                for (int e = 0; e < 8; ++e) { dz[e] = prs * (gdy[e] - s1 - nh[e] * s2); sz[e] += dz[e]; }
                store16(DZ + (r0 + RPP * i) * LDE + 8 * c, dz);
                float x1v[8];
                unpack8(pn1, x1v);
#pragma unroll
                for (int e = 0; e < 8; ++e) x1v[e] = ok ? x1v[e] * g1v[e] + b1v[e] : 0.f;
                store16(X + (r0 + RPP * i) * LDE + 8 * c, x1v);
            }
            f32x4 ca[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) ca[k] = cacc[k * NT + tid];
            __builtin_amdgcn_sched_barrier(0);          // (all six reads in flight before the first add)
            cacc[0 * NT + tid] = ca[0] + f32x4{sg[0], sg[1], sg[2], sg[3]}; cacc[1 * NT + tid] = ca[1] + f32x4{sg[4], sg[5], sg[6], sg[7]};
            cacc[2 * NT + tid] = ca[2] + f32x4{sb[0], sb[1], sb[2], sb[3]}; cacc[3 * NT + tid] = ca[3] + f32x4{sb[4], sb[5], sb[6], sb[7]};
            cacc[4 * NT + tid] = ca[4] + f32x4{sz[0], sz[1], sz[2], sz[3]}; cacc[5 * NT + tid] = ca[5] + f32x4{sz[4], sz[5], sz[6], sz[7]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the staging slots have been read: they may be refilled
        PT_MARK(1);                                             // 1: element-wise phase
        prefetch(min(tile + (int)gridDim.x, ntiles - 1));       // next tile's rows: in flight under phases 1 and 2
        barrier();          // raw s_barrier (LDS drained): __syncthreads() also waits vmcnt(0), i.e. for the row prefetch just issued and for store acks
        PT_MARK(2);                                             // 2: prefetch issue + barrier
        // ---------------- phase 1: u, dh for this wave's 64 hidden units; dW1, dW2 ----------------
        bf16x8 hB[NTW], dhB[NTW];
        {
            // u and dh for 32 of the wave's 64 hidden units at a time (register budget): un-transposed products, so the
            // accumulators (lane = hidden unit, registers = rows) are the row-contraction operands of the weight gradients.
            // (Requesting a step's operand fragments one step ahead -- 16-24 more registers in flight, built in five forms: both row
            // tiles, one row tile at a time with copied or alternating fragment sets, W1 fragments in-step, half-step skew -- makes the
            // compiler spill 11-39 registers of the fc2^T slice at the 512 a wave has; their scratch reloads sit in vmcnt behind the
            // row prefetch and the row stores.  The loop stays un-prefetched: one exposed LDS round trip per 8 MFMAs.)
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            const int dst_lane = (FW * wave + cq) * R + 4 * g, dst_sw0 = (q & 1) << 4, dst_sw1 = ((q & 1) ^ 1) << 4;
#pragma unroll
            for (int hf = 0; hf < NTW / 2; ++hf) {
                f32x4 u[2][2], dh[2][2];
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2) {
                    const float bv = b1f[FW * wave + 32 * hf + 16 * n2 + cq];
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) { u[rt][n2] = f32x4{bv, bv, bv, bv}; dh[rt][n2] = zero; }
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        const bf16x8 xa = lds_frag(X + (16 * rt + cq) * LDE + ks * 32 + 8 * g);
                        const bf16x8 za = lds_frag(DZ + (16 * rt + cq) * LDE + ks * 32 + 8 * g);
#pragma unroll
                        for (int n2 = 0; n2 < 2; ++n2) {
                            const int nt = 2 * hf + n2;
                            u[rt][n2] = mma(xa, lds_frag(W1i + (FW * wave + 16 * nt + cq) * LDE + ks * 32 + 8 * g), u[rt][n2]);
                            dh[rt][n2] = mma(za, w2f[nt][ks], dh[rt][n2]);
                        }
                    }
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2) {
                    const int nt = 2 * hf + n2;
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float uu = u[rt][n2][j];
                            dh[rt][n2][j] *= act_bwd(ACT, ACT == MIVIT_ACT_GELU ? uu : act_fwd(ACT, uu));
                            u[rt][n2][j] = act_fwd(ACT, uu);
                            db1[nt] += dh[rt][n2][j];
                        }
                        // rows 16 rt + 4 g .. + 3 of hidden unit 64 wave + 16 nt + cq: one 8-byte store at dht_off(64 wave + 16 nt + cq,
                        // 16 rt + 4 g), lane-dependent part factored out (this unit's swizzle bit is q & 1) so that (nt, rt) only add
                        // compile-time constants
                        const f32x4 v = dh[rt][n2];
                        const bf16x4 vb = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                        *reinterpret_cast<bf16x4 *>(DH + dst_lane + (rt ? dst_sw1 : dst_sw0) + 16 * nt * R) = vb;
                    }
                    hB[nt] = pack8(u[0][n2], u[1][n2]);
                    dhB[nt] = pack8(dh[0][n2], dh[1][n2]);
                }
            }
        }
        PT_MARK(3);                                             // 3: u, dh
        // weight gradients: the contraction runs over the 32 rows of the tile (slots 0-3 = rows 4g.., slots 4-7 = rows 16+4g..)
        {   // (operands of column tile t+1 are requested before the MFMAs of tile t: one wave per SIMD has nobody else to hide LDS latency)
            bf16x8 xb = tr_pair(X + (4 * g + q) * LDE + 4 * pp, X + (16 + 4 * g + q) * LDE + 4 * pp);
            bf16x8 zb = tr_pair(DZ + (4 * g + q) * LDE + 4 * pp, DZ + (16 + 4 * g + q) * LDE + 4 * pp);
#pragma unroll
            for (int t = 0; t < ET; ++t) {
                bf16x8 xn = xb, zn = zb;
                if (t < ET - 1) {
                    xn = tr_pair(X + (4 * g + q) * LDE + 16 * (t + 1) + 4 * pp, X + (16 + 4 * g + q) * LDE + 16 * (t + 1) + 4 * pp);
                    zn = tr_pair(DZ + (4 * g + q) * LDE + 16 * (t + 1) + 4 * pp, DZ + (16 + 4 * g + q) * LDE + 16 * (t + 1) + 4 * pp);
                }
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    dW1[nt][t] = mma(dhB[nt], xb, dW1[nt][t]);       // [n][k] += dh^T x1
                    dW2[t][nt] = mma(zb, hB[nt], dW2[t][nt]);        // [e][n] += dz2^T h
                }
                xb = xn; zb = zn;
            }
        }
        PT_MARK(4);                                             // 4: dW1, dW2
        barrier();          // raw s_barrier (LDS drained): __syncthreads() also waits vmcnt(0), i.e. for the row prefetch just issued and for store acks
        PT_MARK(5);                                             // 5: barrier
        // ---------------- phase 2: dx1^T for this wave's 32 input features ----------------
        f32x4 dx[KT2][2];
#pragma unroll
        for (int kt = 0; kt < KT2; ++kt)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) dx[kt][rt] = zero;
        // A = W1^T read transposed out of the W1 image: k-slots 0-3 = hidden units 32ks + 4g + {0..3}, slots 4-7 = 32ks + 16 + 4g + {0..3};
        // B = dh^T by the same transposing read out of the [hidden][row] image: identical k-slot order on both operands
        {
            auto wfrag = [&](int ks, int kt) {
                return tr_pair(W1i + (32 * ks + 4 * g + q) * LDE + EW * wave + 16 * kt + 4 * pp,
                               W1i + (32 * ks + 16 + 4 * g + q) * LDE + EW * wave + 16 * kt + 4 * pp);
            };
            // = tr_pair(DH + dht_off(32 ks + 4 g + q, 16 rt + 4 pp), DH + dht_off(32 ks + 16 + 4 g + q, 16 rt + 4 pp)): the swizzle bit of
            // hidden units 32 ks (+ 16) + 4 g + q is g & 1 -- two lane-dependent bases, everything else immediate offsets
            const bf16 *dbase[2] = {DH + (4 * g + q) * R + ((g & 1) << 4) + 4 * pp, DH + (4 * g + q) * R + (((g & 1) ^ 1) << 4) + 4 * pp};
            auto dfrag = [&](int ks, int rt) { return tr_pair(dbase[rt] + 32 * ks * R, dbase[rt] + (32 * ks + 16) * R); };
            bf16x8 d0 = dfrag(0, 0), d1 = dfrag(0, 1);
            bf16x8 wf[KT2];
#pragma unroll
            for (int kt = 0; kt < KT2; ++kt) wf[kt] = wfrag(0, kt);
#pragma unroll
            for (int ks = 0; ks < FS; ++ks) {
                bf16x8 e0 = d0, e1 = d1, vf[KT2];
#pragma unroll
                for (int kt = 0; kt < KT2; ++kt) vf[kt] = wf[kt];
                if (ks < FS - 1) {
                    e0 = dfrag(ks + 1, 0); e1 = dfrag(ks + 1, 1);
#pragma unroll
                    for (int kt = 0; kt < KT2; ++kt) vf[kt] = wfrag(ks + 1, kt);
                }
#pragma unroll
                for (int kt = 0; kt < KT2; ++kt) { dx[kt][0] = mma(wf[kt], d0, dx[kt][0]); dx[kt][1] = mma(wf[kt], d1, dx[kt][1]); }
                d0 = e0; d1 = e1;
#pragma unroll
                for (int kt = 0; kt < KT2; ++kt) wf[kt] = vf[kt];
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int row = row0 + 16 * rt + cq;
#pragma unroll
            for (int kt = 0; kt < KT2; ++kt) {
                const int col = EW * wave + 16 * kt + 4 * g;
                const uint2 zr = *reinterpret_cast<const uint2 *>(DZ + (16 * rt + cq) * LDE + col);
                f32x4 o = dx[kt][rt];
                o[0] += elem_lo(zr.x); o[1] += elem_hi(zr.x);
                o[2] += elem_lo(zr.y); o[3] += elem_hi(zr.y);
                if (row < a.M) {
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                    const bf16x4 ob = {(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
                    *reinterpret_cast<bf16x4 *>(a.dx1 + (int64_t)row * E + col) = ob;
                }
            }
        }
        PT_MARK(6);                                             // 6: dx1 + row stores
        barrier();          // raw s_barrier (LDS drained): __syncthreads() also waits vmcnt(0), i.e. for the row prefetch just issued and for store acks
        PT_MARK(7);                                             // 7: barrier
    }
#ifdef MIVIT_PHASE_TIMING
    if (a.dbg && lane == 0)
        for (int k = 0; k < 8; ++k) a.dbg[((int64_t)blockIdx.x * NW + wave) * 8 + k] = pt[k];
#endif

    // ---------------- partial gradients -> this workgroup's slab ----------------
    float *sl = a.slabs + (int64_t)blockIdx.x * SL_TOTAL;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
        for (int t = 0; t < ET; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sl[SL_W1 + (FW * wave + 16 * nt + 4 * g + j) * E + 16 * t + cq] = dW1[nt][t][j];
                sl[SL_W2 + (16 * t + 4 * g + j) * F + FW * wave + 16 * nt + cq] = dW2[t][nt][j];
            }
        const float s = x4_sum(db1[nt]);
        if (g == 0) sl[SL_B1 + FW * wave + 16 * nt + cq] = s;
    }
    // column sums of the element-wise phase: thread (c, r0) holds columns 8c .. 8c+7; fold the RPP row-threads of a column
    __syncthreads();
    for (int i = tid; i < 3 * E; i += NT) {
        const int which = i / E, col = i - which * E, cc = col >> 3, e = col & 7;
        float s = 0.f;
        for (int rr = 0; rr < RPP; ++rr) s += cacc[(2 * which + (e >> 2)) * NT + cc + CPR * rr][e & 3];
        sl[(which == 0 ? SL_G2 : which == 1 ? SL_BE2 : SL_B2) + col] = s;
    }
}

// ================================================================================================================
// The same block with the hidden units split EIGHT ways (8 waves = two per SIMD, 256 registers each): a wave keeps the
// [32 x 128] + [128 x 32] blocks of dW1 / dW2 of ITS 32 hidden units (128 accumulator registers) and the matching fc2^T slice
// (32 registers), so a second wave fits on every SIMD and covers the other's LDS round trips -- the four-wave kernel above has
// nobody to cover them (3.6 k of its 10.6 k cycles per tile are the un-prefetched u / dh loop, scripts/phase_timing.py).
// The element-wise phase is split by ROLE: waves 0-3 turn dy / n2 into the dz2 image and the three column-sum vectors, waves 4-7
// turn n1 into the x1 image; each wave DMA-prefetches only what its role consumes (6 / 2 pieces per tile).  Phase 2: wave w owns
// input features 16 w .. 16 w + 15 (one 8-byte row store per row tile: the count behind wait_vm<2>).  Same LDS layout, same
// slab layout, same arithmetic per element (dW / db accumulate the same products in the same row order per hidden unit).
// ================================================================================================================
#ifndef MIVIT_WIDTH64          // (width 128 only: at width 64 the four-wave kernel above already runs at 2 workgroups' worth of registers per SIMD)
constexpr int NW8 = 8, NT8 = NW8 * 64, NH = NT8 / 2;         // NH = 256 threads per role

template <int ACT>
__global__ __launch_bounds__(NT8) void mlp_block_bwd8_kernel(const MlpBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16 *W1i = reinterpret_cast<bf16 *>(smem + OFF_W1);
    bf16 *X = reinterpret_cast<bf16 *>(smem + OFF_X), *DZ = reinterpret_cast<bf16 *>(smem + OFF_DZ), *DH = reinterpret_cast<bf16 *>(smem + OFF_DH);
    float *b1f = reinterpret_cast<float *>(smem + OFF_VEC);
    f32x4 *cacc = reinterpret_cast<f32x4 *>(smem + OFF_ACC);
    uint4 *stg = reinterpret_cast<uint4 *>(smem + OFF_STG);
    float *rst = reinterpret_cast<float *>(smem + OFF_RST);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, cq = lane & 15, q = cq >> 2, pp = cq & 3;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);       // (scalar: role tests and DMA destinations stay in SGPRs)
    const bool ew = wave_u < 4;                      // role: element-wise LayerNorm backward (dz2, column sums) | x1 image
    const int th = tid & (NH - 1), wh = wave_u & 3;  // thread / wave index inside the role
    const unsigned stg_a = __builtin_amdgcn_readfirstlane(lds_addr(stg)) + wh * 1024u, rst_a = __builtin_amdgcn_readfirstlane(lds_addr(rst)) + wh * 256u;

    // ---- one-time staging ----
    static_assert((F * (E / 8)) % (4 * NT8) == 0, "W1 staging batches");
    for (int i0 = tid; i0 < F * (E / 8); i0 += 4 * NT8) {
        uint4 v0, v1, v2, v3;
        {
            const int i = i0, n = i / (E / 8), cc = i - n * (E / 8);
            v0 = *reinterpret_cast<const uint4 *>(a.W1 + (int64_t)n * E + cc * 8);
        }
        { const int i = i0 + NT8, n = i / (E / 8), cc = i - n * (E / 8); v1 = *reinterpret_cast<const uint4 *>(a.W1 + (int64_t)n * E + cc * 8); }
        { const int i = i0 + 2 * NT8, n = i / (E / 8), cc = i - n * (E / 8); v2 = *reinterpret_cast<const uint4 *>(a.W1 + (int64_t)n * E + cc * 8); }
        { const int i = i0 + 3 * NT8, n = i / (E / 8), cc = i - n * (E / 8); v3 = *reinterpret_cast<const uint4 *>(a.W1 + (int64_t)n * E + cc * 8); }
        { const int i = i0, n = i / (E / 8), cc = i - n * (E / 8); *reinterpret_cast<uint4 *>(W1i + n * LDE + cc * 8) = v0; }
        { const int i = i0 + NT8, n = i / (E / 8), cc = i - n * (E / 8); *reinterpret_cast<uint4 *>(W1i + n * LDE + cc * 8) = v1; }
        { const int i = i0 + 2 * NT8, n = i / (E / 8), cc = i - n * (E / 8); *reinterpret_cast<uint4 *>(W1i + n * LDE + cc * 8) = v2; }
        { const int i = i0 + 3 * NT8, n = i / (E / 8), cc = i - n * (E / 8); *reinterpret_cast<uint4 *>(W1i + n * LDE + cc * 8) = v3; }
    }
    for (int n = tid; n < F; n += NT8) b1f[n] = a.b1[n];
    for (int i = tid; i < E; i += NT8) { b1f[F + i] = a.gamma2[i]; b1f[F + E + i] = a.gamma1[i]; b1f[F + 2 * E + i] = a.beta1[i]; }
    // fc2^T slice of this wave's hidden units n = 32 wave + 16 nt + cq: w2f[nt][ks] = W2[32 ks + 8 g .. + 7][n]
    bf16x8 w2f[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = 32 * wave + 16 * nt + cq;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = __builtin_bit_cast(__bf16, a.W2[(int64_t)(ks * 32 + 8 * g + e) * F + n].v);
            w2f[nt][ks] = t;
        }
    }
    typedef __attribute__((address_space(3))) const float lds_cf;
    if (ew) {
#pragma unroll
        for (int i = 0; i < 6; ++i) cacc[i * NH + th] = f32x4{0.f, 0.f, 0.f, 0.f};       // (thread-private slots)
    }
    f32x4 dW1[2][8], dW2[8][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { dW1[i][j] = zero; dW2[j][i] = zero; }
    float db1[2] = {0.f, 0.f};
    __syncthreads();

    const int ntiles = (a.M + R - 1) / R;
    int tile = blockIdx.x;
    // staging slots (thread-private, [slot][NH] x 16 B): 0-3 = dy, n2 of the two rows (role ew), 4-5 = n1 of the two rows (role xw)
    auto prefetch = [&](int t, int r0, int c) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // (wave-uniform base + 32-bit byte offset: M * E * 2 < 2^32 is checked by the launcher)
            const int row = min(t * R + r0 + 16 * i, a.M - 1);
            const unsigned o = ((unsigned)row * E + 8 * c) * 2u;
            if (ew) {
                dma16_sbase(a.dy, o, stg_a + (2 * i + 0) * NH * 16u);
                dma16_sbase(a.n2, o, stg_a + (2 * i + 1) * NH * 16u);
                dma4_sbase(a.rstd2, (unsigned)row * 4u, rst_a + i * NH * 4u);
            } else {
                dma16_sbase(a.n1, o, stg_a + (4 + i) * NH * 16u);
            }
        }
    };
    if (tile < ntiles) prefetch(tile, th >> 4, th & 15);
    bool first = true;
#ifdef MIVIT_PHASE_TIMING
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_last = __builtin_amdgcn_s_memtime();
#endif
    for (; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * R;
        // ---------------- phase 0 (by role) ----------------
        // The element-wise slice of a thread (inside its role): columns 8c .. 8c+7 of rows r0 and r0 + 16 of the tile.  Everything
        // addressed from it (staging slots, column-sum slots, image rows, constants) is re-derived from an OPAQUE copy of the
        // thread index every tile: derived once before the loop, the compiler keeps a dozen loop-invariant LDS address registers
        // alive across all three phases -- and at the 256 registers a two-waves-per-SIMD kernel has, spills them (their scratch
        // reloads sit in vmcnt behind the row prefetch).
        int thv = th;
        asm volatile("" : "+v"(thv));
        const int c = thv & 15, r0 = thv >> 4;
        lds_cf *gam2 = (lds_cf *)(b1f + F + 8 * c), *gam1 = (lds_cf *)(b1f + F + E + 8 * c), *bet1 = (lds_cf *)(b1f + F + 2 * E + 8 * c);
        // VM program order of a wave: P(t) [6 (ew) / 2 (xw) DMA pieces: its staged rows of tile t] | the 2 row stores of tile
        // t-1's phase 2 (one 8-byte store per row tile, issued by every wave of a full tile; the one partial tile is the last of
        // the launch) | this wait: vmcnt(2) = P(t) landed.
        if (first) wait_vm<0>(); else wait_vm<2>();
        first = false;
        PT_MARK(0);
        if (ew) {
            float sg[8], sb[8], sz[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) sg[e] = sb[e] = sz[e] = 0.f;
            typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
            const f32x4 g2a = *(lds_cf4 *)(gam2), g2b = *(lds_cf4 *)(gam2 + 4);
            const float g2v[8] = {g2a[0], g2a[1], g2a[2], g2a[3], g2b[0], g2b[1], g2b[2], g2b[3]};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float d[8], nh[8];
                const bool ok = row0 + r0 + 16 * i < a.M;          // rows past the end were clamped to a real row: cancel them
                const uint4 pdy = keep_if(stg[(2 * i + 0) * NH + thv], ok), pn2 = stg[(2 * i + 1) * NH + thv];
                unpack8(pdy, d); unpack8(pn2, nh);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sg[e] += d[e] * nh[e]; sb[e] += d[e];
                    d[e] *= g2v[e];                                // d now holds gamma2 * dy
                    s1 += d[e]; s2 += d[e] * nh[e];
                }
                s1 = g16_sum(s1) * (1.f / E); s2 = g16_sum(s2) * (1.f / E);
                const float prs = __uint_as_float(__float_as_uint(rst[i * NH + thv]) & (ok ? 0xffffffffu : 0u));
                float dz[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { dz[e] = prs * (d[e] - s1 - nh[e] * s2); sz[e] += dz[e]; }
                store16(DZ + (r0 + 16 * i) * LDE + 8 * c, dz);
                __builtin_amdgcn_sched_barrier(0);       // one row at a time (interleaving the two doubles the working set)
            }
            {   // the running column sums, two batches of three 16-byte reads (register budget)
                f32x4 ca[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) ca[k] = cacc[k * NH + thv];
                __builtin_amdgcn_sched_barrier(0);
                cacc[0 * NH + thv] = ca[0] + f32x4{sg[0], sg[1], sg[2], sg[3]}; cacc[1 * NH + thv] = ca[1] + f32x4{sg[4], sg[5], sg[6], sg[7]};
                cacc[2 * NH + thv] = ca[2] + f32x4{sb[0], sb[1], sb[2], sb[3]};
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 3; ++k) ca[k] = cacc[(3 + k) * NH + thv];
                __builtin_amdgcn_sched_barrier(0);
                cacc[3 * NH + thv] = ca[0] + f32x4{sb[4], sb[5], sb[6], sb[7]};
                cacc[4 * NH + thv] = ca[1] + f32x4{sz[0], sz[1], sz[2], sz[3]}; cacc[5 * NH + thv] = ca[2] + f32x4{sz[4], sz[5], sz[6], sz[7]};
            }
        } else {
            typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
            const f32x4 g1a = *(lds_cf4 *)(gam1), g1b = *(lds_cf4 *)(gam1 + 4), b1a = *(lds_cf4 *)(bet1), b1b = *(lds_cf4 *)(bet1 + 4);
            const float g1v[8] = {g1a[0], g1a[1], g1a[2], g1a[3], g1b[0], g1b[1], g1b[2], g1b[3]};
            const float b1v[8] = {b1a[0], b1a[1], b1a[2], b1a[3], b1b[0], b1b[1], b1b[2], b1b[3]};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool ok = row0 + r0 + 16 * i < a.M;
                const uint4 pn1 = keep_if(stg[(4 + i) * NH + thv], ok);
                float x1v[8];
                unpack8(pn1, x1v);
#pragma unroll
                for (int e = 0; e < 8; ++e) x1v[e] = ok ? x1v[e] * g1v[e] + b1v[e] : 0.f;
                store16(X + (r0 + 16 * i) * LDE + 8 * c, x1v);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the staging slots have been read: they may be refilled
        PT_MARK(1);
        prefetch(min(tile + (int)gridDim.x, ntiles - 1), r0, c);       // next tile's rows: in flight under phases 1 and 2
        barrier();
        PT_MARK(2);
        // ---------------- phase 1: u, dh for this wave's 32 hidden units; dW1, dW2 ----------------
        bf16x8 hB[2], dhB[2];
        {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            const int dst_lane = (32 * wave + cq) * R + 4 * g, dst_sw0 = (q & 1) << 4, dst_sw1 = ((q & 1) ^ 1) << 4;
            bf16x4 hlo[2], dlo[2];                  // rows 0-15 of h / dh, packed, while rows 16-31 are computed (register budget)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                f32x4 u[2], dh[2];
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2) {
                    const float bv = b1f[32 * wave + 16 * n2 + cq];
                    u[n2] = f32x4{bv, bv, bv, bv}; dh[n2] = zero;
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 xa = lds_frag(X + (16 * rt + cq) * LDE + ks * 32 + 8 * g);
                    const bf16x8 za = lds_frag(DZ + (16 * rt + cq) * LDE + ks * 32 + 8 * g);
#pragma unroll
                    for (int n2 = 0; n2 < 2; ++n2) {
                        u[n2] = mma(xa, lds_frag(W1i + (32 * wave + 16 * n2 + cq) * LDE + ks * 32 + 8 * g), u[n2]);
                        dh[n2] = mma(za, w2f[n2][ks], dh[n2]);
                    }
                }
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float uu = u[n2][j];
                        dh[n2][j] *= act_bwd(ACT, ACT == MIVIT_ACT_GELU ? uu : act_fwd(ACT, uu));
                        u[n2][j] = act_fwd(ACT, uu);
                        db1[n2] += dh[n2][j];
                    }
                    const bf16x4 hb = {(__bf16)u[n2][0], (__bf16)u[n2][1], (__bf16)u[n2][2], (__bf16)u[n2][3]};
                    const bf16x4 db = {(__bf16)dh[n2][0], (__bf16)dh[n2][1], (__bf16)dh[n2][2], (__bf16)dh[n2][3]};
                    // rows 16 rt + 4 g .. + 3 of hidden unit 32 wave + 16 n2 + cq (dht_off with the lane part factored out:
                    // this unit's swizzle bit is q & 1)
                    *reinterpret_cast<bf16x4 *>(DH + dst_lane + (rt ? dst_sw1 : dst_sw0) + 16 * n2 * R) = db;
                    if (rt == 0) { hlo[n2] = hb; dlo[n2] = db; }
                    else {
                        struct { bf16x4 lo, hi; } ph = {hlo[n2], hb}, pd = {dlo[n2], db};
                        hB[n2] = __builtin_bit_cast(bf16x8, ph);
                        dhB[n2] = __builtin_bit_cast(bf16x8, pd);
                    }
                }
            }
        }
        PT_MARK(3);
        {
            bf16x8 xb = tr_pair(X + (4 * g + q) * LDE + 4 * pp, X + (16 + 4 * g + q) * LDE + 4 * pp);
            bf16x8 zb = tr_pair(DZ + (4 * g + q) * LDE + 4 * pp, DZ + (16 + 4 * g + q) * LDE + 4 * pp);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                bf16x8 xn = xb, zn = zb;
                if (t < 7) {
                    xn = tr_pair(X + (4 * g + q) * LDE + 16 * (t + 1) + 4 * pp, X + (16 + 4 * g + q) * LDE + 16 * (t + 1) + 4 * pp);
                    zn = tr_pair(DZ + (4 * g + q) * LDE + 16 * (t + 1) + 4 * pp, DZ + (16 + 4 * g + q) * LDE + 16 * (t + 1) + 4 * pp);
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    dW1[nt][t] = mma(dhB[nt], xb, dW1[nt][t]);       // [n][k] += dh^T x1
                    dW2[t][nt] = mma(zb, hB[nt], dW2[t][nt]);        // [e][n] += dz2^T h
                }
                xb = xn; zb = zn;
            }
        }
        PT_MARK(4);
        barrier();
        PT_MARK(5);
        // ---------------- phase 2: dx1^T for this wave's 16 input features ----------------
        f32x4 dx[2] = {zero, zero};
        {
            auto wfrag = [&](int ks) {
                return tr_pair(W1i + (32 * ks + 4 * g + q) * LDE + 16 * wave + 4 * pp, W1i + (32 * ks + 16 + 4 * g + q) * LDE + 16 * wave + 4 * pp);
            };
            const bf16 *dbase[2] = {DH + (4 * g + q) * R + ((g & 1) << 4) + 4 * pp, DH + (4 * g + q) * R + (((g & 1) ^ 1) << 4) + 4 * pp};
            auto dfrag = [&](int ks, int rt) { return tr_pair(dbase[rt] + 32 * ks * R, dbase[rt] + (32 * ks + 16) * R); };
            bf16x8 d0 = dfrag(0, 0), d1 = dfrag(0, 1), w0 = wfrag(0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                bf16x8 e0 = d0, e1 = d1, v0 = w0;
                if (ks < 7) { e0 = dfrag(ks + 1, 0); e1 = dfrag(ks + 1, 1); v0 = wfrag(ks + 1); }
                dx[0] = mma(w0, d0, dx[0]); dx[1] = mma(w0, d1, dx[1]);
                d0 = e0; d1 = e1; w0 = v0;
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int row = row0 + 16 * rt + cq, col = 16 * wave + 4 * g;
            const uint2 zr = *reinterpret_cast<const uint2 *>(DZ + (16 * rt + cq) * LDE + col);
            f32x4 o = dx[rt];
            o[0] += elem_lo(zr.x); o[1] += elem_hi(zr.x);
            o[2] += elem_lo(zr.y); o[3] += elem_hi(zr.y);
            if (row < a.M) {
                typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                const bf16x4 ob = {(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
                *reinterpret_cast<bf16x4 *>(a.dx1 + (int64_t)row * E + col) = ob;
            }
        }
        PT_MARK(6);
        barrier();
        PT_MARK(7);
    }
#ifdef MIVIT_PHASE_TIMING
    if (a.dbg && lane == 0)
        for (int kk = 0; kk < 8; ++kk) a.dbg[((int64_t)blockIdx.x * NW8 + wave) * 8 + kk] = pt[kk];
#endif

    // ---------------- partial gradients -> this workgroup's slab ----------------
    float *sl = a.slabs + (int64_t)blockIdx.x * SL_TOTAL;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sl[SL_W1 + (32 * wave + 16 * nt + 4 * g + j) * E + 16 * t + cq] = dW1[nt][t][j];
                sl[SL_W2 + (16 * t + 4 * g + j) * F + 32 * wave + 16 * nt + cq] = dW2[t][nt][j];
            }
        const float sv = x4_sum(db1[nt]);
        if (g == 0) sl[SL_B1 + 32 * wave + 16 * nt + cq] = sv;
    }
    __syncthreads();
    for (int i = tid; i < 3 * E; i += NT8) {
        const int which = i / E, col = i - which * E, cc = col >> 3, e = col & 7;
        float sv = 0.f;
        for (int rr = 0; rr < 16; ++rr) sv += cacc[(2 * which + (e >> 2)) * NH + cc + 16 * rr][e & 3];
        sl[(which == 0 ? SL_G2 : which == 1 ? SL_BE2 : SL_B2) + col] = sv;
    }
}

#endif          // !MIVIT_WIDTH64

// ================================================================================================================
// LayerNorm-1 backward + out-projection backward (autograd of x1 = LN1(x + out_proj(ctx)), models.py:57,100-102, up to the
// attention core):   in : dy = dL/dx1, n1 / rstd1 (LN1's normalised output, 1/std), ctx (the out-projection's input)
//                    out: dz1 = dL/d(pre-norm sum) [the residual branch's gradient], dctx = dz1 Wo, dWo = dz1^T ctx, dbo, dgamma1, dbeta1
// one pass: 3 row reads, 2 row writes (the three launches it replaces -- LayerNorm backward, weight gradient, data gradient --
// moved 7).  Same structure as the feed-forward backward, without its recompute: 32-row tiles, element-wise LayerNorm
// backward by all threads into an LDS image, then wave w takes output features 32w..32w+31 of dctx (transposed product:
// 8-byte row stores) and rows 32w..32w+31 of dWo (row contraction: both operands by transposing LDS reads).  Small LDS and
// register footprint -> two workgroups per CU cover each other's barriers.
// ================================================================================================================
constexpr int AO_OFF_DZ = 0, AO_OFF_CT = AO_OFF_DZ + R * LDE * 2, AO_OFF_STG = AO_OFF_CT + R * LDE * 2;
constexpr int AO_OFF_RED = AO_OFF_STG + 3 * RPT * NT * 16;      // [NW][3][E] floats: end-of-launch fold of the column sums
constexpr int AO_OFF_RST = AO_OFF_RED + NW * 3 * E * 4;        // [RPT][NT] floats: rstd1 of the next tile's rows
constexpr int AO_LDS = AO_OFF_RST + RPT * NT * 4;
constexpr int AO_SL_W = 0, AO_SL_B = E * E, AO_SL_G = AO_SL_B + E, AO_SL_BE = AO_SL_G + E, AO_SL_TOTAL = AO_SL_BE + E;   // arena order:
                                                                // out_proj.weight, out_proj.bias, norm1.weight, norm1.bias
struct AttnOutBwdArgs {
    const bf16 *dy, *n1; const float *rstd1, *gamma1;
    const bf16 *ctx, *Wo;
    int M;
    bf16 *dz1, *dctx;
    float *slabs;
};

__global__ __launch_bounds__(NT, 2) void attn_out_bwd_kernel(const AttnOutBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16 *DZ = reinterpret_cast<bf16 *>(smem + AO_OFF_DZ), *CT = reinterpret_cast<bf16 *>(smem + AO_OFF_CT);
    uint4 *stg = reinterpret_cast<uint4 *>(smem + AO_OFF_STG);
    float *red = reinterpret_cast<float *>(smem + AO_OFF_RED);
    float *rst = reinterpret_cast<float *>(smem + AO_OFF_RST);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, cq = lane & 15, q = cq >> 2, pp = cq & 3;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // row operand of dctx^T = Wo^T dz1^T for this wave's context features c = 32 wave + 16 ct + cq: A[c][e] = Wo[e][c]
    bf16x8 wof[KT2][KS];
#pragma unroll
    for (int ct = 0; ct < KT2; ++ct)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = __builtin_bit_cast(__bf16, a.Wo[(int64_t)(ks * 32 + 8 * g + e) * E + EW * wave + 16 * ct + cq].v);
            wof[ct][ks] = t;
        }
    const int c = tid % CPR, r0 = tid / CPR;
    float gam[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gam[e] = a.gamma1[8 * c + e];
    float sg[8], sb[8], sz[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sg[e] = sb[e] = sz[e] = 0.f;
    f32x4 dWo[KT2][ET];
#pragma unroll
    for (int i = 0; i < KT2; ++i)
#pragma unroll
        for (int j = 0; j < ET; ++j) dWo[i][j] = zero;

    const int ntiles = (a.M + R - 1) / R;
    int tile = blockIdx.x;
    auto prefetch = [&](int t) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int64_t row = min((int64_t)t * R + r0 + RPP * i, (int64_t)a.M - 1);
            const int64_t o = row * E + 8 * c;
            dma16_opaque(a.dy + o, stg + (3 * i + 0) * NT + wave * 64);
            dma16_opaque(a.n1 + o, stg + (3 * i + 1) * NT + wave * 64);
            dma16_opaque(a.ctx + o, stg + (3 * i + 2) * NT + wave * 64);
            dma4_opaque(a.rstd1 + row, rst + i * NT + wave * 64);
        }
    };
    if (tile < ntiles) prefetch(tile);
    bool first = true;
    for (; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = (int64_t)tile * R;
        // ---- phase 0: LayerNorm backward, dz1 -> HBM + LDS image, ctx -> LDS image ----
        // the staged rows were requested one tile ago; younger than them are only the four dctx row stores of the last
        // phase 1 (every full tile issues them; the one partial tile is the last of the launch): do not wait for their acks
        // (width 64: NROWST = 2 of them)
        if (first) wait_vm<0>(); else wait_vm<NROWST>();
        first = false;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            float d[8], nh[8], gdy[8];
            const bool ok = row0 + r0 + RPP * i < a.M;
            const uint4 pdy = keep_if(stg[(3 * i + 0) * NT + tid], ok), pn = stg[(3 * i + 1) * NT + tid];
            const uint4 pct = keep_if(stg[(3 * i + 2) * NT + tid], ok);
            unpack8(pdy, d); unpack8(pn, nh);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                gdy[e] = d[e] * gam[e];
                s1 += gdy[e]; s2 += gdy[e] * nh[e];
                sg[e] += d[e] * nh[e]; sb[e] += d[e];
            }
            s1 = g16_sum(s1) * (1.f / E); s2 = g16_sum(s2) * (1.f / E);
            float dz[8];
            const float prs = __uint_as_float(__float_as_uint(rst[i * NT + tid]) & (ok ? 0xffffffffu : 0u));
#pragma unroll
            for (int e = 0; e < 8; ++e) { dz[e] = prs * (gdy[e] - s1 - nh[e] * s2); sz[e] += dz[e]; }
            store16(DZ + (r0 + RPP * i) * LDE + 8 * c, dz);
            if (ok) store16(a.dz1 + (row0 + r0 + RPP * i) * E + 8 * c, dz);
            *reinterpret_cast<uint4 *>(CT + (r0 + RPP * i) * LDE + 8 * c) = pct;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        prefetch(min(tile + (int)gridDim.x, ntiles - 1));
        barrier();          // raw s_barrier (LDS drained): __syncthreads() also waits vmcnt(0), i.e. for the row prefetch just issued and for store acks
        // ---- phase 1: dctx^T for this wave's 32 context features; dWo rows 32 wave .. +31 ----
        f32x4 dc[KT2][2];
#pragma unroll
        for (int ct = 0; ct < KT2; ++ct) { dc[ct][0] = zero; dc[ct][1] = zero; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 z0 = lds_frag(DZ + cq * LDE + ks * 32 + 8 * g), z1 = lds_frag(DZ + (16 + cq) * LDE + ks * 32 + 8 * g);
#pragma unroll
            for (int ct = 0; ct < KT2; ++ct) { dc[ct][0] = mma(wof[ct][ks], z0, dc[ct][0]); dc[ct][1] = mma(wof[ct][ks], z1, dc[ct][1]); }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int64_t row = row0 + 16 * rt + cq;
            if (row < a.M) {
#pragma unroll
                for (int ct = 0; ct < KT2; ++ct) {
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                    const f32x4 o = dc[ct][rt];
                    const bf16x4 ob = {(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
                    *reinterpret_cast<bf16x4 *>(a.dctx + row * E + EW * wave + 16 * ct + 4 * g) = ob;
                }
            }
        }
#pragma unroll
        for (int et = 0; et < KT2; ++et) {
            const bf16x8 za = tr_pair(DZ + (4 * g + q) * LDE + EW * wave + 16 * et + 4 * pp, DZ + (16 + 4 * g + q) * LDE + EW * wave + 16 * et + 4 * pp);
#pragma unroll
            for (int ct = 0; ct < ET; ++ct) {
                const bf16x8 cb = tr_pair(CT + (4 * g + q) * LDE + 16 * ct + 4 * pp, CT + (16 + 4 * g + q) * LDE + 16 * ct + 4 * pp);
                dWo[et][ct] = mma(za, cb, dWo[et][ct]);                 // [e][c] += dz1^T ctx
            }
        }
        barrier();          // raw s_barrier (LDS drained): __syncthreads() also waits vmcnt(0), i.e. for the row prefetch just issued and for store acks
    }
    // ---- partial gradients -> this workgroup's slab ----
    float *sl = a.slabs + (int64_t)blockIdx.x * AO_SL_TOTAL;
#pragma unroll
    for (int et = 0; et < KT2; ++et)
#pragma unroll
        for (int ct = 0; ct < ET; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) sl[AO_SL_W + (EW * wave + 16 * et + 4 * g + j) * E + 16 * ct + cq] = dWo[et][ct][j];
    // column sums: the wave's row-threads of chunk c are the lanes with the same lane % CPR -- the four 16-lane rows, and at
    // width 64 both halves of each (row_ror:8)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float vg = x4_sum(sg[e]), vb = x4_sum(sb[e]), vz = x4_sum(sz[e]);
        if constexpr (CPR == 8) { vg += dpp_mov<0x128>(vg); vb += dpp_mov<0x128>(vb); vz += dpp_mov<0x128>(vz); }
        if (g == 0 && cq < CPR) { red[(wave * 3 + 0) * E + 8 * c + e] = vg; red[(wave * 3 + 1) * E + 8 * c + e] = vb; red[(wave * 3 + 2) * E + 8 * c + e] = vz; }
    }
    __syncthreads();
    for (int i = tid; i < 3 * E; i += NT) {
        const int which = i / E, col = i - which * E;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[(w * 3 + which) * E + col];
        sl[(which == 0 ? AO_SL_G : which == 1 ? AO_SL_BE : AO_SL_B) + col] = v;
    }
}

// ================================================================================================================
// q|k|v projection backward: weight gradient AND data gradient in one pass over dqkv (autograd of qkv = x Wqkv^T + bqkv,
// reference helpers/models.py:42-44, plus the residual branch's gradient that joins d(x) there, :100-102):
//   in : dqkv [M, 3E] (the attention core's backward), x [M, E] (the projection's input rows as the engine holds them: the
//        producing LayerNorm's normalised output, or the token rows for layer 0), Wqkv [3E, E], res [M, E] = d(x) of the residual
//   out: dx = dqkv Wqkv + res [M, E];  dW = dqkv^T x [3E, E], db = column sums of dqkv [3E]  (fp32, overwritten)
// dqkv is read ONCE (the two launches this replaces -- wgrad_dma and the 384-deep row-stream data gradient -- read it twice:
// 9 U -> 6 U of traffic per layer).  Same skeleton as attn_out_bwd: 32-row tiles requested one tile ahead by LDS-DMA into
// thread-private slots, copied into padded operand images, then wave w (E / 16 of them: 8 | 4) takes
//   * features 16 w .. 16 w + 15 of dx^T = Wqkv^T dqkv^T (its slice of Wqkv^T lives in registers; + res; 8-byte row stores), and
//   * rows 48 w .. 48 w + 47 of dW (row contraction over the tile: both operands by transposing LDS reads) and of db (the same
//     dqkv^T fragments against a vector of ones: one more MFMA per fragment, no element-wise pass, no fold).
// Per-workgroup partial gradients go to slabs, reduced in a fixed order (deterministic).
// ================================================================================================================
constexpr int QN = 3 * E, QCH = QN / 8;                  // q|k|v width; its 16-byte chunks per row (48 | 24)
constexpr int NWQ = E / 16, NTQ = NWQ * 64;              // waves, threads per workgroup
constexpr int KQ = QN / 32;                              // contraction steps of the data gradient (12 | 6)
constexpr int LDQ = QN + 16;                             // image pitch: 800 | 416 B = 2 | 10 mod 16 units, 8 | 40 banks per row
constexpr int QB_OFF_DQ = 0, QB_OFF_X = QB_OFF_DQ + R * LDQ * 2, QB_OFF_RES = QB_OFF_X + R * LDE * 2, QB_OFF_STG = QB_OFF_RES + R * LDE * 2;
constexpr int QB_LDS = QB_OFF_STG + 5 * NTQ * 16;        // staging: 3 dqkv chunks + 1 x chunk + 1 res chunk per thread and tile
constexpr int QB_SL_W = 0, QB_SL_B = QN * E, QB_SL_TOTAL = QB_SL_B + QN;          // arena order: in_proj weight, bias
constexpr int QB_WGPC = E == 64 ? 2 : 1;                 // resident workgroups per CU the grid is sized for
static_assert(R * QCH == 3 * NTQ && R * CPR == NTQ && QN == 48 * NWQ, "q|k|v backward: thread / wave slices");
static_assert(QB_LDS * QB_WGPC <= 160 * 1024, "LDS budget");

struct QkvBwdArgs {
    const bf16 *dqkv, *x, *W, *res;
    int M;
    bf16 *dx;
    float *slabs;          // [gridDim.x][QB_SL_TOTAL]
    // optional: x is the normalised output of a LayerNorm whose affine (gamma, beta) belongs to the projection's true input
    // gamma * x + beta: the weight gradient against that input is dW * diag(gamma) + db (x) beta -- linear in (dW, db), so it is
    // applied to this workgroup's partial sums on their way into the slab (it was a launch of its own behind the reduction)
    const float *fix_gamma, *fix_beta;
};

__global__ __launch_bounds__(NTQ) void qkv_bwd_kernel(const QkvBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16 *DQ = reinterpret_cast<bf16 *>(smem + QB_OFF_DQ), *X = reinterpret_cast<bf16 *>(smem + QB_OFF_X), *RS = reinterpret_cast<bf16 *>(smem + QB_OFF_RES);
    uint4 *stg = reinterpret_cast<uint4 *>(smem + QB_OFF_STG);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, cq = lane & 15, q = cq >> 2, pp = cq & 3;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // row operand of dx^T = Wqkv^T dqkv^T for this wave's features c = 16 wave + cq: A[c][n] = Wqkv[n][c]
    bf16x8 wof[KQ];
#pragma unroll
    for (int ks = 0; ks < KQ; ++ks) {
        bf16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = __builtin_bit_cast(__bf16, a.W[(int64_t)(ks * 32 + 8 * g + e) * E + 16 * wave + cq].v);
        wof[ks] = t;
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    f32x4 dW[3][ET], dbv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        dbv[i] = zero;
#pragma unroll
        for (int j = 0; j < ET; ++j) dW[i][j] = zero;
    }
    // this thread's slices of a tile: dqkv chunks i = tid + NTQ j (row i / QCH, chunk i % QCH), one chunk (row r0, chunk c) of x and res
    const int c = tid % CPR, r0 = tid / CPR;
    const int ntiles = (a.M + R - 1) / R;
    int tile = blockIdx.x;
    auto prefetch = [&](int t) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i = tid + NTQ * j, rr = i / QCH, cc = i - rr * QCH;
            const int64_t row = min((int64_t)t * R + rr, (int64_t)a.M - 1);
            dma16_opaque(a.dqkv + row * QN + 8 * cc, stg + j * NTQ + wave * 64);
        }
        const int64_t row = min((int64_t)t * R + r0, (int64_t)a.M - 1);
        dma16_opaque(a.x + row * E + 8 * c, stg + 3 * NTQ + wave * 64);
        dma16_opaque(a.res + row * E + 8 * c, stg + 4 * NTQ + wave * 64);
    };
    if (tile < ntiles) prefetch(tile);
    bool first = true;
    for (; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = (int64_t)tile * R;
        // ---- phase 0: staged chunks -> operand images (rows past the end were clamped to a real row: zeroed here) ----
        // VM program order of a wave: P(t) [5 DMA pieces] | the 2 dx row stores of tile t-1 (one per row tile: every full tile
        // issues them; the one partial tile is the last of the launch) | this wait: vmcnt(2) = P(t) landed.
        if (first) wait_vm<0>(); else wait_vm<2>();
        first = false;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i = tid + NTQ * j, rr = i / QCH, cc = i - rr * QCH;
            *reinterpret_cast<uint4 *>(DQ + rr * LDQ + 8 * cc) = keep_if(stg[j * NTQ + tid], row0 + rr < a.M);
        }
        *reinterpret_cast<uint4 *>(X + r0 * LDE + 8 * c) = keep_if(stg[3 * NTQ + tid], row0 + r0 < a.M);
        *reinterpret_cast<uint4 *>(RS + r0 * LDE + 8 * c) = stg[4 * NTQ + tid];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the staging slots have been read: they may be refilled
        prefetch(min(tile + (int)gridDim.x, ntiles - 1));
        barrier();          // raw s_barrier (LDS drained): __syncthreads() would also wait for the prefetch just issued and for store acks
        // ---- phase 1: dx^T for this wave's 16 features ----
        f32x4 dc[2] = {zero, zero};
#pragma unroll
        for (int ks = 0; ks < KQ; ++ks) {
            const bf16x8 z0 = lds_frag(DQ + cq * LDQ + ks * 32 + 8 * g), z1 = lds_frag(DQ + (16 + cq) * LDQ + ks * 32 + 8 * g);
            dc[0] = mma(wof[ks], z0, dc[0]);
            dc[1] = mma(wof[ks], z1, dc[1]);
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int64_t row = row0 + 16 * rt + cq;
            const uint2 rr = *reinterpret_cast<const uint2 *>(RS + (16 * rt + cq) * LDE + 16 * wave + 4 * g);
            f32x4 o = dc[rt];
            o[0] += elem_lo(rr.x); o[1] += elem_hi(rr.x);
            o[2] += elem_lo(rr.y); o[3] += elem_hi(rr.y);
            if (row < a.M) {
                typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                const bf16x4 ob = {(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
                *reinterpret_cast<bf16x4 *>(a.dx + row * E + 16 * wave + 4 * g) = ob;
            }
        }
        // ---- phase 2: dW rows 48 wave .. + 47 (and db): the contraction runs over the 32 rows of the tile ----
        bf16x8 za[3];
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            za[nt] = tr_pair(DQ + (4 * g + q) * LDQ + 48 * wave + 16 * nt + 4 * pp, DQ + (16 + 4 * g + q) * LDQ + 48 * wave + 16 * nt + 4 * pp);
            dbv[nt] = mma(za[nt], ones, dbv[nt]);                  // [n][*] += sum over rows of dqkv[row][n]
        }
#pragma unroll
        for (int ct = 0; ct < ET; ++ct) {
            const bf16x8 cb = tr_pair(X + (4 * g + q) * LDE + 16 * ct + 4 * pp, X + (16 + 4 * g + q) * LDE + 16 * ct + 4 * pp);
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) dW[nt][ct] = mma(za[nt], cb, dW[nt][ct]);       // [n][k] += dqkv^T x
        }
        barrier();
    }
    // ---- partial gradients -> this workgroup's slab ----
    float *sl = a.slabs + (int64_t)blockIdx.x * QB_SL_TOTAL;
    const bool fix = a.fix_gamma != nullptr;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
#pragma unroll
        for (int ct = 0; ct < ET; ++ct) {
            const float gk = fix ? a.fix_gamma[16 * ct + cq] : 1.f, bk = fix ? a.fix_beta[16 * ct + cq] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)          // (dbv[nt][j]: every column of the ones-product holds the row's sum)
                sl[QB_SL_W + (48 * wave + 16 * nt + 4 * g + j) * E + 16 * ct + cq] = fix ? dW[nt][ct][j] * gk + dbv[nt][j] * bk : dW[nt][ct][j];
        }
        if (cq == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sl[QB_SL_B + 48 * wave + 16 * nt + 4 * g + j] = dbv[nt][j];
        }
    }
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
int grid_for(int M) { return std::min(256, ceil_div(M, R)); }
int qkv_grid_for(int M) { return std::min(256 * QB_WGPC, ceil_div(M, R)); }

}  // namespace

size_t qkv_bwd_ws_bytes(int M) { return align_up((size_t)qkv_grid_for(std::max(M, 1)) * QB_SL_TOTAL * sizeof(float), 256); }

// dx [M,E]; dW [3E,E], db [3E] fp32 (overwritten); fix_gamma / fix_beta [E] optional (see QkvBwdArgs)
int launch_qkv_bwd(const void *dqkv, const void *x, const void *Wqkv, const void *res, int M, void *dx, float *dW, float *db,
                   const float *fix_gamma, const float *fix_beta, void *ws, size_t ws_bytes, hipStream_t s) {
    MIVIT_CHECK(dqkv && x && Wqkv && res && dx && dW && db && ws && M > 0, "qkv_bwd: null pointer / empty problem");
    MIVIT_CHECK(aligned16(dqkv) && aligned16(x) && aligned16(res) && aligned16(dx), "qkv_bwd: pointers must be 16-byte aligned");
    MIVIT_CHECK(ws_bytes >= qkv_bwd_ws_bytes(M), "qkv_bwd: workspace too small");
    QkvBwdArgs a{static_cast<const bf16 *>(dqkv), static_cast<const bf16 *>(x), static_cast<const bf16 *>(Wqkv), static_cast<const bf16 *>(res),
                 M, static_cast<bf16 *>(dx), static_cast<float *>(ws), fix_gamma, fix_gamma ? fix_beta : nullptr};
    MIVIT_CHECK((fix_gamma == nullptr) == (fix_beta == nullptr), "qkv_bwd: the input affine needs both gamma and beta");
    const int grid = qkv_grid_for(M);
    {
        ProfScope prof(s);
        MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(qkv_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, QB_LDS));
        hipLaunchKernelGGL(qkv_bwd_kernel, dim3(grid), dim3(NTQ), QB_LDS, s, a);
        MIVIT_LAUNCH_CHECK();
    }
    const float *sl = static_cast<const float *>(ws);
    if (db == dW + QB_SL_B) return launch_slab_reduce_strided(sl, grid, QB_SL_TOTAL, QB_SL_TOTAL, dW, s);
    RC(launch_slab_reduce_strided(sl + QB_SL_W, grid, QB_SL_TOTAL, QN * E, dW, s));
    return launch_slab_reduce_strided(sl + QB_SL_B, grid, QB_SL_TOTAL, QN, db, s);
}

#ifdef MIVIT_WIDTH64
static const int g_mlp_bwd_waves = 4;
constexpr int NW8 = 8;
#else
static int g_mlp_bwd_waves = [] { const char *e = getenv("MIVIT_MLP_BWD_WAVES"); return e && atoi(e) == 4 ? 4 : 8; }();
#endif
#if !defined(MIVIT_ELEM_F16) && !defined(MIVIT_WIDTH64)
extern "C" int mivit_mlp_block_bwd_set_waves(int waves) {
    const int old = g_mlp_bwd_waves;
    if (waves == 4 || waves == 8) g_mlp_bwd_waves = waves;
    return old;
}
#endif

size_t mlp_block_bwd_ws_bytes(int M) { return align_up((size_t)grid_for(std::max(M, 1)) * SL_TOTAL * sizeof(float), 256); }

// dW1 [F,E], db1 [F], dW2 [E,F], db2 [E], dgamma2, dbeta2 [E]: overwritten
int launch_mlp_block_bwd(const void *dy, const void *n2, const float *rstd2, const float *gamma2, const void *n1,
                         const float *gamma1, const float *beta1, const void *W1, const float *b1, const void *W2, int M, int act,
                         void *dx1, float *dW1, float *db1, float *dW2, float *db2, float *dgamma2, float *dbeta2, void *ws,
                         size_t ws_bytes, hipStream_t s) {
    MIVIT_CHECK(dy && n2 && rstd2 && gamma2 && n1 && gamma1 && beta1 && W1 && b1 && W2 && dx1 && dW1 && db1 && dW2 && db2 &&
                dgamma2 && dbeta2 && ws && M > 0, "mlp_block_bwd: null pointer / empty problem");
    MIVIT_CHECK(aligned16(dy) && aligned16(n2) && aligned16(n1) && aligned16(W1) && aligned16(dx1), "mlp_block_bwd: pointers must be 16-byte aligned");
    MIVIT_CHECK(ws_bytes >= mlp_block_bwd_ws_bytes(M), "mlp_block_bwd: workspace too small");
    MlpBwdArgs a{static_cast<const bf16 *>(dy), static_cast<const bf16 *>(n2), rstd2, gamma2, static_cast<const bf16 *>(n1), gamma1,
                 beta1, static_cast<const bf16 *>(W1), b1, static_cast<const bf16 *>(W2), M, static_cast<bf16 *>(dx1),
                 static_cast<float *>(ws), nullptr};
    const int grid = grid_for(M);
    // MIVIT_MLP_BWD_WAVES / mivit_mlp_block_bwd_set_waves: 4 = the first kernel (hidden units split four ways, one wave per SIMD);
    // 8 (default) = split eight ways
    const int waves = g_mlp_bwd_waves;
    MIVIT_CHECK((int64_t)M * E * 2 < (1ll << 32), "mlp_block_bwd: %d rows exceed the 32-bit byte offsets of the row prefetch", M);
#ifdef MIVIT_PHASE_TIMING
    static unsigned long long *dbg_buf = nullptr;
    if (!dbg_buf) MIVIT_HIP(hipMalloc(&dbg_buf, 256 * NW8 * 8 * sizeof(unsigned long long)));
    a.dbg = dbg_buf;
#endif
    {
        ProfScope prof(s);
#ifdef MIVIT_WIDTH64
#define BWD_LAUNCH8(ACT_) do { } while (0)
#else
#define BWD_LAUNCH8(ACT_)                                                                                        \
    do {                                                                                                         \
        auto kern = mlp_block_bwd8_kernel<ACT_>;                                                                 \
        MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT8), LDS_BYTES, s, a);                                        \
    } while (0)
#endif
#define BWD_LAUNCH(ACT_)                                                                                         \
    do {                                                                                                         \
        if (waves == 8) {                                                                                        \
            BWD_LAUNCH8(ACT_);                                                                                   \
        } else {                                                                                                 \
            auto kern = mlp_block_bwd_kernel<ACT_>;                                                              \
            MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)); \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), LDS_BYTES, s, a);                                     \
        }                                                                                                        \
    } while (0)
        switch (act) {
            case MIVIT_ACT_RELU: BWD_LAUNCH(MIVIT_ACT_RELU); break;
            case MIVIT_ACT_LEAKY_RELU: BWD_LAUNCH(MIVIT_ACT_LEAKY_RELU); break;
            case MIVIT_ACT_GELU: BWD_LAUNCH(MIVIT_ACT_GELU); break;
            default: BWD_LAUNCH(MIVIT_ACT_NONE); break;
        }
#undef BWD_LAUNCH
#undef BWD_LAUNCH8
        MIVIT_LAUNCH_CHECK();
    }
#ifdef MIVIT_PHASE_TIMING
    {
        static const char *names[8] = {"wait rows", "element-wise", "prefetch+barrier", "u, dh", "dW1, dW2", "barrier", "dx1 + stores", "barrier"};
        const int nwv = waves == 8 ? NW8 : NW;
        std::vector<unsigned long long> h((size_t)grid * nwv * 8);
        MIVIT_HIP(hipStreamSynchronize(s));
        MIVIT_HIP(hipMemcpy(h.data(), dbg_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        const int tiles = ceil_div(M, R);
        double tot = 0;
        for (int k = 0; k < 8; ++k) {
            double sum = 0;
            double sum_ew = 0, sum_xw = 0;
            for (int i = 0; i < grid * nwv; ++i) {
                sum += (double)h[(size_t)i * 8 + k];
                ((i % nwv) < 4 ? sum_ew : sum_xw) += (double)h[(size_t)i * 8 + k];
            }
            if (nwv == 8) fprintf(stderr, "[mlp_block_bwd phases]   waves 0-3 %8.1f   waves 4-7 %8.1f\n", sum_ew / 4 / tiles, sum_xw / 4 / tiles);
            const double per_tile = sum / nwv / tiles;                       // shader-clock cycles per tile of one workgroup, wave average
            tot += per_tile;
            fprintf(stderr, "[mlp_block_bwd phases] %-18s %8.1f cycles/tile\n", names[k], per_tile);
        }
        fprintf(stderr, "[mlp_block_bwd phases] total %8.1f cycles/tile, %d tiles, grid %d\n", tot, tiles, grid);
    }
#endif
    const float *sl = static_cast<const float *>(ws);
    if (db1 == dW1 + SL_B1 && dW2 == dW1 + SL_W2 && db2 == dW1 + SL_B2 && dgamma2 == dW1 + SL_G2 && dbeta2 == dW1 + SL_BE2)
        return launch_slab_reduce_strided(sl, grid, SL_TOTAL, SL_TOTAL, dW1, s);          // the arena's layout: one launch
    struct { int off, n; float *out; } parts[] = {{SL_W1, F * E, dW1}, {SL_W2, E * F, dW2}, {SL_B1, F, db1}, {SL_B2, E, db2},
                                                  {SL_G2, E, dgamma2}, {SL_BE2, E, dbeta2}};
    for (auto &p : parts) RC(launch_slab_reduce_strided(sl + p.off, grid, SL_TOTAL, p.n, p.out, s));
    return 0;
}

#ifndef MIVIT_ELEM_F16      // operator-level C-ABI: declared for bf16 (include/mivit_hip.h; the width-64 build exports ..._w64)
extern "C" size_t mivit_mlp_block_bwd_workspace_bytes(int M) { return mlp_block_bwd_ws_bytes(M); }
extern "C" int mivit_mlp_block_bwd(const void *dy, const void *n2, const float *rstd2, const float *gamma2, const void *n1,
                                   const float *gamma1, const float *beta1, const void *W1_bf16, const float *b1,
                                   const void *W2_bf16, int M, int act, void *dx1, float *dW1, float *db1, float *dW2, float *db2,
                                   float *dgamma2, float *dbeta2, void *workspace, size_t workspace_bytes, void *stream) {
    prof_set_tag(MIVIT_PROF_OP);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return launch_mlp_block_bwd(dy, n2, rstd2, gamma2, n1, gamma1, beta1, W1_bf16, b1, W2_bf16, M, act, dx1, dW1, db1, dW2, db2, dgamma2,
                                dbeta2, workspace, workspace_bytes, s);
}

#endif
size_t attn_out_bwd_ws_bytes(int M) { return align_up((size_t)std::min(512, ceil_div(std::max(M, 1), R)) * AO_SL_TOTAL * sizeof(float), 256); }

// dz1, dctx [M,E] bf16; dWo [E,E], dbo, dgamma1, dbeta1 [E] fp32 (overwritten)
int launch_attn_out_bwd(const void *dy, const void *n1, const float *rstd1, const float *gamma1, const void *ctx, const void *Wo, int M,
                        void *dz1, void *dctx, float *dWo, float *dbo, float *dgamma1, float *dbeta1, void *ws, size_t ws_bytes,
                        hipStream_t s) {
    MIVIT_CHECK(dy && n1 && rstd1 && gamma1 && ctx && Wo && dz1 && dctx && dWo && dbo && dgamma1 && dbeta1 && ws && M > 0,
                "attn_out_bwd: null pointer / empty problem");
    MIVIT_CHECK(aligned16(dy) && aligned16(n1) && aligned16(ctx) && aligned16(Wo) && aligned16(dz1) && aligned16(dctx),
                "attn_out_bwd: pointers must be 16-byte aligned");
    MIVIT_CHECK(ws_bytes >= attn_out_bwd_ws_bytes(M), "attn_out_bwd: workspace too small");
    AttnOutBwdArgs a{static_cast<const bf16 *>(dy), static_cast<const bf16 *>(n1), rstd1, gamma1, static_cast<const bf16 *>(ctx),
                     static_cast<const bf16 *>(Wo), M, static_cast<bf16 *>(dz1), static_cast<bf16 *>(dctx), static_cast<float *>(ws)};
    const int grid = std::min(512, ceil_div(M, R));            // two resident workgroups per CU
    {
        ProfScope prof(s);
        MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(attn_out_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, AO_LDS));
        hipLaunchKernelGGL(attn_out_bwd_kernel, dim3(grid), dim3(NT), AO_LDS, s, a);
        MIVIT_LAUNCH_CHECK();
    }
    const float *sl = static_cast<const float *>(ws);
    if (dbo == dWo + AO_SL_B && dgamma1 == dWo + AO_SL_G && dbeta1 == dWo + AO_SL_BE)
        return launch_slab_reduce_strided(sl, grid, AO_SL_TOTAL, AO_SL_TOTAL, dWo, s);
    struct { int off, n; float *out; } parts[] = {{AO_SL_W, E * E, dWo}, {AO_SL_B, E, dbo}, {AO_SL_G, E, dgamma1}, {AO_SL_BE, E, dbeta1}};
    for (auto &p : parts) RC(launch_slab_reduce_strided(sl + p.off, grid, AO_SL_TOTAL, p.n, p.out, s));
    return 0;
}

#ifndef MIVIT_ELEM_F16
extern "C" size_t mivit_qkv_bwd_workspace_bytes(int M) { return qkv_bwd_ws_bytes(M); }
extern "C" int mivit_qkv_bwd(const void *dqkv, const void *x, const void *Wqkv_bf16, const void *res, int M, void *dx, float *dW, float *db,
                             void *workspace, size_t workspace_bytes, void *stream) {
    prof_set_tag(MIVIT_PROF_OP);
    return launch_qkv_bwd(dqkv, x, Wqkv_bf16, res, M, dx, dW, db, nullptr, nullptr, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}
extern "C" size_t mivit_attn_out_bwd_workspace_bytes(int M) { return attn_out_bwd_ws_bytes(M); }
extern "C" int mivit_attn_out_bwd(const void *dy, const void *n1, const float *rstd1, const float *gamma1, const void *ctx,
                                  const void *Wo_bf16, int M, void *dz1, void *dctx, float *dWo, float *dbo, float *dgamma1,
                                  float *dbeta1, void *workspace, size_t workspace_bytes, void *stream) {
    prof_set_tag(MIVIT_PROF_OP);
    return launch_attn_out_bwd(dy, n1, rstd1, gamma1, ctx, Wo_bf16, M, dz1, dctx, dWo, dbo, dgamma1, dbeta1, workspace, workspace_bytes,
                               static_cast<hipStream_t>(stream));
}
#endif
