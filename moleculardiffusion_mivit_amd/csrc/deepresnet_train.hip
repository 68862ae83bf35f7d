// DeepResNetEmbedding, TRAINING mode (batch-statistics BatchNorm) -- reference helpers/models.py:230-257 (+ ResidualBlock
// :202-228), forward and backward, hand-written for gfx950.
//
// Data layout: every activation is a pixel-major matrix [N*P*P, C] (NHWC) of the compute type T (fp32 / bf16) in HBM.
// The RAW convolution outputs y_i are what is stored; BatchNorm + ReLU (and the residual add) are never materialised:
// every consumer applies them while it copies its input tile into LDS ("normalise on load"):
//      ACT1: a  = relu(scale*y + shift)                          ACT2: o = relu(scale*y + shift + scale'*y' + shift')
//      DY  : dy = k*g + c0 + c1*y   (BatchNorm backward folded into three per-channel coefficients; g = masked upstream
//                                    gradient, written once by the mask/reduce kernel)
// A convolution is an implicit GEMM on MFMA over F whole frames that sit in LDS as zero-haloed [pixel][channel] images
// (rows = output pixels, k = (tap, input channel)); weights are read from an L2-resident [c_out][tap][c_in] pack.  The
// same kernel serves forward (epilogue: raw output + per-channel sum / sum-of-squares partials for the batch
// statistics) and data-gradient (flipped-tap pack, DY prologue; the 1x1 skip branch's gradient is a second pass that
// starts its accumulators from the 3x3 pass's output).  LDS layouts are bank-conflict-free by construction (Img).
// The weight gradient contracts over pixels: both operands are read TRANSPOSED out of their natural LDS images with
// ds_read_tr16_b64 (bf16) / scalar reads (fp32); partial sums per workgroup go to a slab that is reduced
// deterministically (no atomics anywhere).
#include "common.h"
#include "stream_prims.h"
#include <algorithm>
#include <type_traits>
#include <vector>

namespace {

#ifdef MIVIT_DRN_PHASE_TIMING
#define DRN_DBG(a) ((a).dbg)
#else
#define DRN_DBG(a) 0
#endif
constexpr int NT = 512, MAXM = 22;          // 8 waves; <= 352 output pixels (22 MFMA row tiles) per workgroup
constexpr int CSTR = 128;                   // stride of the per-channel coefficient tables
template <typename T> constexpr int vec_el() { return 16 / (int)sizeof(T); }

const int DRN_CO[7] = {32, 64, 64, 64, 128, 128, 128};
const int DRN_CI[7] = {1, 32, 64, 32, 64, 128, 64};
const int DRN_TAPS[7] = {9, 9, 9, 1, 9, 9, 1};
// conv / BatchNorm index: 0 initial, 1 block1.conv1, 2 block1.conv2, 3 block1.skip, 4 block2.conv1, 5 block2.conv2, 6 block2.skip

template <typename T>
__device__ __forceinline__ typename Mma<T>::Frag frag_at(const T *p);
template <>
__device__ __forceinline__ float frag_at<float>(const float *p) { return *p; }
template <>
__device__ __forceinline__ bf16x8 frag_at<bf16>(const bf16 *p) { return *reinterpret_cast<const bf16x8 *>(p); }

// Work decomposition.  A frame is cut into tiles of th x tw output pixels (one tile = the whole frame when it is small
// enough); a workgroup holds F tile "slots", each a zero-haloed (th+2) x (tw+2) image in LDS.  Halo cells that fall
// inside the frame are loaded from the neighbouring pixels (the activations live in HBM between layers), cells outside
// the frame are the convolution's zero padding.  Row r of the workgroup = (slot r / (th*tw), ty, tx).
struct Geom {
    int N, P, PP;
    int th, tw, ntx, tpf, units;     // tile size, tiles per frame row, tiles per frame, N * tpf
    int F, HPt, RT;                  // slots per workgroup, cells per slot image, F * th * tw rows (<= MAXM * 16 = 352)
};

// per-workgroup lookup tables in LDS: rowg[r] = global pixel row (frame*P*P + y*P + x) or -1, rowc[r] = halo cell of row r
__device__ __forceinline__ void build_row_tables(const Geom &g, int group, int nrows, int *rowg, int *rowc, int tid) {
    const int tt = g.th * g.tw, W2 = g.tw + 2;
    for (int r = tid; r < nrows; r += 512) {
        int gr = -1, cell = W2 + 1;                        // dead rows alias the first interior cell of slot 0
        const int s = r / tt;
        if (s < g.F) {
            const int u = group * g.F + s, rem = r - s * tt, ty = rem / g.tw, tx = rem - ty * g.tw;
            const int frame = u / g.tpf, ti = u - frame * g.tpf, y = (ti / g.ntx) * g.th + ty, x = (ti % g.ntx) * g.tw + tx;
            cell = s * g.HPt + (ty + 1) * W2 + tx + 1;
            if (u < g.units && y < g.P && x < g.P) gr = frame * g.PP + y * g.P + x;
        }
        rowg[r] = gr; rowc[r] = cell;
    }
}

// global pixel row behind image cell `cell` of this workgroup, or -1 (zero padding / dead slot)
__device__ __forceinline__ int cell_source(const Geom &g, int group, int cell) {
    const int W2 = g.tw + 2, s = cell / g.HPt, rem = cell - s * g.HPt, cy = rem / W2, cx = rem - cy * W2;
    const int u = group * g.F + s;
    if (u >= g.units) return -1;
    const int frame = u / g.tpf, ti = u - frame * g.tpf, y = (ti / g.ntx) * g.th + cy - 1, x = (ti % g.ntx) * g.tw + cx - 1;
    if (y < 0 || y >= g.P || x < 0 || x >= g.P) return -1;
    return frame * g.PP + y * g.P + x;
}

// ---------------------------------------------------------------------------------------------------------------
// normalise-on-load
// ---------------------------------------------------------------------------------------------------------------
enum { PRO_ACT1 = 0, PRO_ACT2 = 1, PRO_DY = 2 };
struct TileSrc {
    const void *p0;     // ACT: y            DY: g
    const void *p1;     // ACT2: y'          DY: y
    const float *ca;    // ACT: forward table of BN(y)  [mean | rstd | scale | shift]     DY: backward table [k | c0 | c1]
    const float *cb;    // ACT2: forward table of BN(y')
};


template <typename T, int CW, int CTOT, int PRO, int U, typename PosFn>
__device__ __forceinline__ void fill_batched(T *tile, const TileSrc &s, int n, const int *srcrow, PosFn pos, int c_off, int tid);

// weight-gradient kernel (accumulators live across the fill: 4 entries in flight): plain row-ordered image [nrows][CB] of
// the channel window c_off .. c_off + CB (dead rows zero); rows 32 B (16-bit) / 16 B (fp32) apart modulo the bank period
template <typename T, int CB, int CTOT, int PRO>
__device__ __forceinline__ void fill_rows(T *tile, const TileSrc &s, const int *rowg, int nrows, int c_off, int tid) {
    constexpr int CS = CB + (sizeof(T) == 2 ? 16 : 4);
    fill_batched<T, CB, CTOT, PRO, 4>(tile, s, nrows, rowg, [](int r) { return r * CS; }, c_off, tid);
}

// ---------------------------------------------------------------------------------------------------------------
// LDS image layout of the convolution kernels (16-bit types): conflict-free A-fragment reads.
// A fragment read is one ds_read_b128 per lane: lane (cq, g) takes the 16 B k-chunk g of output pixel cq's cell.  The
// instruction is served in groups of 16 lanes that mix two k-chunks, so with a cell stride of s 16-byte units the group is
// conflict-free only for s = 2, 6, 10, 14 (mod 16): 32 B of padding per cell for 32 / 64 / 128 channels (16 B made every
// group 2-way).  Consecutive output pixels must also look like consecutive cells to the banks, so each image row is padded
// to cancel the two halo cells a row wrap skips, and each slot to cancel the jump to the next frame (measured before:
// SQ_LDS_BANK_CONFLICT = 40-54 % of SQ_LDS_IDX_ACTIVE on every convolution; model: 2.1-2.9 -> 1.05 cycles per group).
// fp32 (parity mode; ds_read_b32 fragments) keeps the plain layout.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int C>
struct Img {
    static constexpr int PADE = sizeof(T) == 2 ? 16 : 4;                    // elements of padding per cell
    static constexpr int CS = C + PADE, CSB = CS * (int)sizeof(T);
    static constexpr int RPB = sizeof(T) == 2 ? (256 - (2 * CSB) % 256) % 256 : 0;     // row padding, bytes
    int rowe, slote;                                                         // elements per image row / per slot
    __host__ __device__ Img(int th, int tw) {
        rowe = (tw + 2) * CS + RPB / (int)sizeof(T);
        const int rowb = rowe * (int)sizeof(T);
        const int spb = sizeof(T) == 2 ? (((th * tw * CSB - (th + 2) * rowb) % 256) + 256) % 256 : 0;
        slote = (th + 2) * rowe + spb / (int)sizeof(T);
    }
};

// cellsrc[cell] = global pixel row behind the cell (or -1), cellpos[cell] = element offset of the cell in an Img layout
__device__ __forceinline__ void build_cell_tables(const Geom &g, int group, int *cellsrc, int *cellpos, int cs, int rowe, int slote,
                                                  int tid, bool interior_only = false) {
    const int W2 = g.tw + 2;
    for (int i = tid; i < g.F * g.HPt; i += 512) {
        const int s = i / g.HPt, rem = i - s * g.HPt, cy = rem / W2, cx = rem - cy * W2;
        const bool halo = cy == 0 || cy == g.th + 1 || cx == 0 || cx == g.tw + 1;
        cellsrc[i] = interior_only && halo ? -1 : cell_source(g, group, i);      // (a 1x1 convolution never reads the halo)
        cellpos[i] = s * slote + cy * rowe + cx * cs;
    }
}

template <typename T>
__device__ __forceinline__ void unpack16(const uint4 &v, float *out) {
    if constexpr (sizeof(T) == 4) {
        out[0] = __uint_as_float(v.x); out[1] = __uint_as_float(v.y); out[2] = __uint_as_float(v.z); out[3] = __uint_as_float(v.w);
    } else {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { out[2 * i] = __uint_as_float(w[i] << 16); out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
}

// Fill an LDS image: entry i (a cell or a row) takes the channel window [c_off, c_off + CW) of row srcrow[i] of a
// CTOT-channel tensor, normalised on the way, and lands at element pos(i); srcrow[i] < 0 stores zero.  A thread's channel
// chunk is the same for all its entries (NT is a multiple of the chunks per entry), so the BatchNorm coefficients sit in
// registers, and the entries are taken U at a time with all their global loads issued before the first use: one entry
// per iteration left every load's HBM latency exposed (the loop was ~2/3 of a convolution workgroup's life).  Zero
// entries load row 0 (cached).
template <typename T, int CW, int CTOT, int PRO, int U, typename PosFn>
__device__ __forceinline__ void fill_batched(T *tile, const TileSrc &s, int n, const int *srcrow, PosFn pos, int c_off, int tid) {
    constexpr int V = vec_el<T>(), CV = CW / V, EPP = NT / CV;          // entries per pass of the workgroup
    static_assert(NT % CV == 0, "fixed channel chunk per thread");
    const int cw = (tid % CV) * V, c = c_off + cw;
    float k0[V], k1[V], k2[V], k3[V];        // (same expressions, in the same order, as the mask and pooling kernels)
#pragma unroll
    for (int e = 0; e < V; ++e) {
        k2[e] = k3[e] = 0.f;
        if (PRO == PRO_ACT1) { k0[e] = s.ca[2 * CSTR + c + e]; k1[e] = s.ca[3 * CSTR + c + e]; }
        else if (PRO == PRO_ACT2) { k0[e] = s.ca[2 * CSTR + c + e]; k1[e] = s.ca[3 * CSTR + c + e]; k2[e] = s.cb[2 * CSTR + c + e]; k3[e] = s.cb[3 * CSTR + c + e]; }
        else { k0[e] = s.ca[c + e]; k1[e] = s.ca[CSTR + c + e]; k2[e] = s.ca[2 * CSTR + c + e]; }
    }
    const T *p0 = static_cast<const T *>(s.p0) + c, *p1 = static_cast<const T *>(s.p1) + c;
    for (int base = tid / CV; base < n; base += U * EPP) {
        int src[U], dst[U];
        uint4 ra[U], rb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * EPP, ic = i < n ? i : base;
            src[u] = srcrow[ic]; dst[u] = i < n ? pos(ic) : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = src[u] > 0 ? src[u] : 0;
            ra[u] = *reinterpret_cast<const uint4 *>(p0 + row * CTOT);
            if (PRO != PRO_ACT1) rb[u] = *reinterpret_cast<const uint4 *>(p1 + row * CTOT);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float a[V], b[V], o[V];
            unpack16<T>(ra[u], a);
            if (PRO != PRO_ACT1) unpack16<T>(rb[u], b);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float v;
                if (PRO == PRO_ACT1) v = fmaxf(k0[e] * a[e] + k1[e], 0.f);
                else if (PRO == PRO_ACT2) v = fmaxf(k0[e] * a[e] + k1[e] + k2[e] * b[e] + k3[e], 0.f);
                else v = k0[e] * a[e] + k1[e] + k2[e] * b[e];
                o[e] = src[u] >= 0 ? v : 0.f;
            }
            if (dst[u] >= 0) store16(tile + dst[u] + cw, o);
        }
    }
}

// zero-haloed Img image of all F slots (convolution kernels)
template <typename T, int C, int PRO, int U = 8>
__device__ __forceinline__ void fill_image(T *tile, const TileSrc &s, const Geom &g, const int *cellsrc, const int *cellpos, int tid) {
    fill_batched<T, C, C, PRO, U>(tile, s, g.F * g.HPt, cellsrc, [&](int cell) { return cellpos[cell]; }, 0, tid);
}

// ---------------------------------------------------------------------------------------------------------------
// implicit-GEMM core: acc[j][mt] += sum_{tap, ci} in[pixel(mt) + tap][ci] * W[co(j)][tap][ci]   (two column tiles / wave)
// weight fragments are prefetched one chunk (G k-steps) ahead: they come from L2, the image fragments from LDS
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int CIN, int TAPS>
struct ConvK {
    static constexpr int KS = Mma<T>::KS, KL = sizeof(T) == 2 ? 8 : 1, CS = Img<T, CIN>::CS;
    static constexpr int NC = CIN / KS, TOT = TAPS * NC, G = 2, NCH = TOT / G;
    static_assert(CIN % KS == 0, "channel count must be a multiple of the MFMA k step");
    static __device__ __forceinline__ int offset(int s, int rowe) {
        const int tap = s / NC, c0 = (s - tap * NC) * KS;
        return (TAPS == 9 ? ((tap / 3 - 1) * rowe + (tap % 3 - 1) * CS) : 0) + c0;
    }
};

// GE k-steps with the weight fragments in `cur` (PREFETCH: the next chunk's fragments are loaded into `nxt` first).
// NTW = 16-column tiles per wave.  A k-step is walked in SUB sub-steps of TS row tiles (NTW = 1 owns twice the row tiles, so
// it takes them in two halves to keep the fragment double buffer at 2 x 11); the image fragments of sub-step s + 1 are read
// between the MFMAs of sub-step s.
template <typename T, int CIN, int TAPS, int NTW, int MT, int MTC, int GE, bool PREFETCH>
__device__ __forceinline__ void conv_chunk(f32x4 (&acc)[NTW][MT], const T *in, const T *const (&w)[NTW], const int (&hidx)[MT], int rowe,
                                           int ch, const typename Mma<T>::Frag (&cur)[NTW][2], typename Mma<T>::Frag (&nxt)[NTW][2]) {
    typedef typename Mma<T>::Frag Frag;
    using K = ConvK<T, CIN, TAPS>;
    constexpr int SUB = NTW == 1 ? 2 : 1, TS = (MTC + SUB - 1) / SUB, NSS = GE * SUB;
    if constexpr (PREFETCH) {
#pragma unroll
        for (int j = 0; j < K::G; ++j) {                   // (clamped: the last prefetch re-reads the final k-step)
            int s = (ch + 1) * K::G + j;
            s = s < K::TOT ? s : K::TOT - 1;
#pragma unroll
            for (int n = 0; n < NTW; ++n) nxt[n][j] = frag_at<T>(w[n] + s * K::KS);
        }
    }
    int off[GE];
#pragma unroll
    for (int j = 0; j < GE; ++j) off[j] = K::offset(ch * K::G + j, rowe);
    Frag a[2][TS];
#pragma unroll
    for (int t = 0; t < TS; ++t) a[0][t] = frag_at<T>(in + hidx[t] + off[0]);
#pragma unroll
    for (int ss = 0; ss < NSS; ++ss) {
        const int j = ss / SUB, base = (ss % SUB) * TS, cnt = MTC - base < TS ? MTC - base : TS;
        const int jn = (ss + 1) / SUB, basen = ((ss + 1) % SUB) * TS, cntn = ss + 1 < NSS ? (MTC - basen < TS ? MTC - basen : TS) : 0;
#pragma unroll
        for (int t = 0; t < TS; ++t) {
            if (t < cnt) {
#pragma unroll
                for (int n = 0; n < NTW; ++n) acc[n][base + t] = Mma<T>::mma(cur[n][j], a[ss & 1][t], acc[n][base + t]);
            }
            if (t < cntn) a[(ss + 1) & 1][t] = frag_at<T>(in + hidx[basen + t] + off[jn < GE ? jn : GE - 1]);
        }
    }
    if constexpr (PREFETCH) __builtin_amdgcn_sched_group_barrier(0x020, NTW * K::G, 0);   // the weight fragments leave first
    __builtin_amdgcn_sched_group_barrier(0x100, TS, 0);
#pragma unroll
    for (int ss = 0; ss < NSS; ++ss) {
        const int base = (ss % SUB) * TS, cnt = MTC - base < TS ? MTC - base : TS;
        const int basen = ((ss + 1) % SUB) * TS, cntn = ss + 1 < NSS ? (MTC - basen < TS ? MTC - basen : TS) : 0;
#pragma unroll
        for (int t = 0; t < TS; ++t) {
            if (t < cnt) __builtin_amdgcn_sched_group_barrier(0x008, NTW, 0);
            if (t < cntn) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }
}

// The first MTC row tiles are computed unconditionally (dead ones read a valid cell and are ignored by the epilogue), so the
// loop body has no branch, and the schedule is pinned with sched_group_barrier: left alone the scheduler serialises
// read -> wait -> 2 MFMAs per tile to save registers (SQ_WAIT_ANY 0.68, MFMA busy 0.23 on the 128 -> 128 convolution).
// Per chunk of two k-steps: the next chunk's weight fragments (from L2) leave first, then all image fragment reads of the
// first sub-step, then each sub-step's MFMAs with the next one's reads slotted in between them.  The two weight-fragment
// buffers swap roles chunk by chunk (no register copies, so a fragment is waited for at its first use one chunk later).
template <typename T, int CIN, int TAPS, int NTW, int MT, int MTC>
__device__ __forceinline__ void conv_accum_t(f32x4 (&acc)[NTW][MT], const T *in, const T *const (&w_)[NTW],
                                             const int (&hidx)[MT], int rowe, int lane) {
    // hidx[mt] = element offset of this lane's output pixel in the Img layout; rowe = elements per image row
    typedef typename Mma<T>::Frag Frag;
    using K = ConvK<T, CIN, TAPS>;
    const int g = lane >> 4;
    const T *w[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) w[n] = w_[n] + g * K::KL;
    in += g * K::KL;
    Frag fa[NTW][K::G], fb[NTW][K::G];
#pragma unroll
    for (int j = 0; j < K::G; ++j) {
        const int s = j < K::TOT ? j : K::TOT - 1;
#pragma unroll
        for (int n = 0; n < NTW; ++n) fa[n][j] = frag_at<T>(w[n] + s * K::KS);
    }
    int ch = 0;
#pragma unroll 1
    for (; ch + 1 < K::NCH; ch += 2) {
        conv_chunk<T, CIN, TAPS, NTW, MT, MTC, K::G, true>(acc, in, w, hidx, rowe, ch, fa, fb);
        conv_chunk<T, CIN, TAPS, NTW, MT, MTC, K::G, true>(acc, in, w, hidx, rowe, ch + 1, fb, fa);
    }
    if constexpr (K::NCH % 2 != 0) conv_chunk<T, CIN, TAPS, NTW, MT, MTC, K::G, true>(acc, in, w, hidx, rowe, K::NCH - 1, fa, fb);
    if constexpr (K::TOT % K::G != 0) {                      // odd number of k-steps: the last one alone
        if constexpr (K::NCH % 2 != 0) conv_chunk<T, CIN, TAPS, NTW, MT, MTC, 1, false>(acc, in, w, hidx, rowe, K::NCH, fb, fa);
        else conv_chunk<T, CIN, TAPS, NTW, MT, MTC, 1, false>(acc, in, w, hidx, rowe, K::NCH, fa, fb);
    }
}

template <typename T, int CIN, int TAPS, int NTW, int MT>
__device__ __forceinline__ void conv_accum(f32x4 (&acc)[NTW][MT], const T *in, const T *const (&w)[NTW],
                                           const int (&hidx)[MT], int nm, int rowe, int lane) {
    constexpr int MH = (MT + 1) / 2;
    if (nm <= 0) return;
    if (nm > MH) conv_accum_t<T, CIN, TAPS, NTW, MT, MT>(acc, in, w, hidx, rowe, lane);       // all tiles (dead ones wasted)
    else conv_accum_t<T, CIN, TAPS, NTW, MT, MH>(acc, in, w, hidx, rowe, lane);              // the first half (small groups)
}

struct ConvArgs {
    Geom g;
    TileSrc A;
    const void *W, *W2;          // [COUT][9][CIN], [COUT][CIN or CIN2]
    void *out, *out2;            // [N*P*P, COUT]
    float *stats, *stats2;       // [blocks][2][COUT] partial sum / sum of squares (EPI 1) or [blocks][3][COUT] mask sums (EPI 2, 3)
    const void *my, *my2;        // EPI 2 / 3: raw outputs y (and y' of the skip branch) of the layer whose ReLU masks `out`
    const float *mca, *mcb;      //            and their forward BatchNorm tables
    int dbg;                     // timing experiments only (MIVIT_DRN_DBG): 1 skip fill, 2 skip MFMA loop, 4 skip epilogue, 8 skip tables
};

// 4 consecutive channels of one pixel <-> one 8-byte (16-bit types) / 16-byte (fp32) access
template <typename T>
__device__ __forceinline__ void store4(T *p, const float *v) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        const uint32_t lo = (uint32_t)from_f32<T>(v[0]).v | ((uint32_t)from_f32<T>(v[1]).v << 16);
        const uint32_t hi = (uint32_t)from_f32<T>(v[2]).v | ((uint32_t)from_f32<T>(v[3]).v << 16);
        *reinterpret_cast<uint2 *>(p) = make_uint2(lo, hi);
    }
}
template <typename T>
__device__ __forceinline__ void load4(const T *p, float *v) {
    if constexpr (sizeof(T) == 4) {
        const float4 f = *reinterpret_cast<const float4 *>(p);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    } else {
        const uint2 u = *reinterpret_cast<const uint2 *>(p);
        v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    }
}

// The accumulators hold the TRANSPOSED product (weights are the MFMA's row operand): acc[j][mt][r] = output pixel
// (mt0 + mt) * 16 + cq, channel (2 ng + j) * 16 + 4 g + r -- four consecutive channels of one pixel per lane, so a row tile
// leaves as one 8-byte store per lane (pixel-major accumulators needed 4 two-byte stores: 169 of the 128 -> 128
// convolution's 1000 us).
// EPI: 0 store | 1 store + per-channel sum / sum of squares partials (forward BatchNorm statistics) | 2 / 3 the ReLU mask of
// the layer below folded in (3: that layer adds a skip branch): g = v * [scale*y + shift (+ scale'*y' + shift') > 0] is what
// is stored, with the partials sum g, sum g*y (, sum g*y') its BatchNorm backward needs -- the separate mask pass over the
// gradient (read v, read y, write g) is gone for every gradient that a convolution produces.
template <typename T, int COUT, int NTW, int MT, int EPI>
__device__ __forceinline__ void conv_epilogue(const f32x4 (&acc)[NTW][MT], T *out, float *stats, float *red, const int *rowg,
                                              int mt0, int nm, int ng, int lane, int wave, int tid, const ConvArgs &a) {
    constexpr int NG = COUT / (16 * NTW), MQ = 8 / NG, NS_ = EPI >= 2 ? 3 : 2;
    const int g = lane >> 4, cq = lane & 15;
    int rows[MT];                                  // this lane's output pixel per row tile: one batch of LDS reads
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rows[mt] = mt < nm ? rowg[(mt0 + mt) * 16 + cq] : -1;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int co = (NTW * ng + j) * 16 + 4 * g;
        float s[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f}, s3[4] = {0.f, 0.f, 0.f, 0.f};
        float sA[4], hA[4], sB[4], hB[4];
        if constexpr (EPI >= 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sA[r] = a.mca[2 * CSTR + co + r]; hA[r] = a.mca[3 * CSTR + co + r];
                sB[r] = EPI == 3 ? a.mcb[2 * CSTR + co + r] : 0.f; hB[r] = EPI == 3 ? a.mcb[3 * CSTR + co + r] : 0.f;
            }
        }
        if constexpr (EPI >= 2) {
            // y (and y') of this lane's pixels, 8 row tiles at a time with all loads issued before the first use
            typedef typename std::conditional<sizeof(T) == 4, float4, uint2>::type Raw;
            constexpr int UB = 8;
#pragma unroll
            for (int mb = 0; mb < MT; mb += UB) {
                Raw yr[UB], y2r[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (mb + u < MT) {
                        const size_t row = rows[mb + u] > 0 ? rows[mb + u] : 0;
                        yr[u] = *reinterpret_cast<const Raw *>(static_cast<const T *>(a.my) + row * COUT + co);
                        if (EPI == 3) y2r[u] = *reinterpret_cast<const Raw *>(static_cast<const T *>(a.my2) + row * COUT + co);
                    }
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (mb + u < MT) {
                        const int mt = mb + u;
                        if (rows[mt] >= 0) {
                            float v[4] = {acc[j][mt][0], acc[j][mt][1], acc[j][mt][2], acc[j][mt][3]};
                            float yv[4], y2v[4] = {0.f, 0.f, 0.f, 0.f};
                            load4<T>(reinterpret_cast<const T *>(&yr[u]), yv);
                            if (EPI == 3) load4<T>(reinterpret_cast<const T *>(&y2r[u]), y2v);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float act = sA[r] * yv[r] + hA[r];              // (the mask kernel's expression, in its order)
                                if (EPI == 3) act += sB[r] * y2v[r] + hB[r];
                                v[r] = act > 0.f ? v[r] : 0.f;
                                s[r] += v[r]; ss[r] += v[r] * yv[r];
                                if (EPI == 3) s3[r] += v[r] * y2v[r];
                            }
                            store4<T>(out + (size_t)rows[mt] * COUT + co, v);
                        }
                    }
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                if (rows[mt] >= 0) {
                    const float v[4] = {acc[j][mt][0], acc[j][mt][1], acc[j][mt][2], acc[j][mt][3]};
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s[r] += v[r]; ss[r] += v[r] * v[r]; }
                    store4<T>(out + (size_t)rows[mt] * COUT + co, v);
                }
        }
        if constexpr (EPI >= 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    s[r] += __shfl_xor(s[r], m, 64); ss[r] += __shfl_xor(ss[r], m, 64);
                    if (EPI == 3) s3[r] += __shfl_xor(s3[r], m, 64);
                }
                if (cq == 0) {
                    red[(wave * NS_ + 0) * 32 + j * 16 + 4 * g + r] = s[r]; red[(wave * NS_ + 1) * 32 + j * 16 + 4 * g + r] = ss[r];
                    if (EPI >= 2) red[(wave * NS_ + 2) * 32 + j * 16 + 4 * g + r] = s3[r];
                }
            }
        }
    }
    if constexpr (EPI >= 1) {
        __syncthreads();
        for (int t = tid; t < NS_ * COUT; t += NT) {
            const int which = t / COUT, c = t - which * COUT, ngc = c / (16 * NTW), cc = c % (16 * NTW);
            float v = 0.f;
#pragma unroll
            for (int mq = 0; mq < MQ; ++mq) v += red[((mq * NG + ngc) * NS_ + which) * 32 + cc];
            stats[which * COUT + c] = v;
        }
        __syncthreads();
    }
}


// SECOND: 0 none | 1 second OUTPUT out2 = conv1x1(tile A, W2) (forward skip branch) | 3 no 3x3 part:
// out += conv1x1(tile A, W2) (the skip's data gradient as its own pass, after the 3x3 pass wrote out; a variant that held
// both gradient images in LDS fitted half the frames per workgroup and was slower than the two passes together)
// MM: row tiles per workgroup (MAXM: one workgroup per CU with up to 160 KB of LDS)
// NTW: 16-column tiles per wave.  The 8 waves are NG = COUT / (16 NTW) column groups x MQ = 8 / NG row groups, and the MQ
// waves of a column group stream the same weight fragments from L2; timing with the weight loads removed showed those loads
// are what the loop waits for (128 -> 128: -18 %, 128 -> 64: -36 %, 64 -> 64: -46 %), so NTW = 1 (half the redundancy, twice the
// LDS image reads per wave, which have room) is the default.
template <typename T, int CIN, int COUT, int SECOND, int PRO, int EPI, int MM, int NTW>
__global__ __launch_bounds__(NT, MM == MAXM ? 2 : 4) void drn_conv_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ROWS_PAD = MM * 16;
    constexpr int NG = COUT / (16 * NTW), MQ = 8 / NG, MT = (MM + MQ - 1) / MQ;
    static_assert(NG >= 1 && NG <= 8 && NG * MQ == 8, "wave grid");
    static_assert(SECOND == 0 || SECOND == 1 || SECOND == 3, "see above");
    static_assert(EPI >= 0 && EPI <= 3 && (SECOND != 1 || EPI == 1), "epilogue kind");
    constexpr int NSTAT = EPI >= 2 ? 3 : 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cq = lane & 15;
    const Geom &g = a.g;
    const int NM = (g.RT + 15) / 16;
    const Img<T, CIN> img(g.th, g.tw);
    T *tileA = reinterpret_cast<T *>(smem);
    float *red = reinterpret_cast<float *>(tileA + g.F * img.slote);   // [8][3][32]
    int *rowg = reinterpret_cast<int *>(red + 8 * 3 * 32), *rowc = rowg + ROWS_PAD, *cellsrc = rowc + ROWS_PAD;
    int *cellpos = cellsrc + g.F * g.HPt;

    build_row_tables(g, blockIdx.x, ROWS_PAD, rowg, rowc, tid);
    build_cell_tables(g, blockIdx.x, cellsrc, cellpos, Img<T, CIN>::CS, img.rowe, img.slote, tid, SECOND == 3);
    __syncthreads();
    if (!(DRN_DBG(a) & 1)) fill_image<T, CIN, PRO, (MM == MAXM ? 8 : 4)>(tileA, a.A, g, cellsrc, cellpos, tid);
    __syncthreads();

    const int ng = wave % NG, mq = wave / NG, per = (NM + MQ - 1) / MQ, mt0 = mq * per;
    const int nm = max(0, min(per, NM - mt0));
    int hidx[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) hidx[mt] = cellpos[rowc[min((mt0 + mt) * 16 + cq, ROWS_PAD - 1)]];
    f32x4 acc[NTW][MT];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[j][mt] = zero;
    const T *w9[NTW], *w1[NTW];                    // this lane's rows of the 3x3 / 1x1 weight packs
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const size_t co = (size_t)((NTW * ng + j) * 16 + cq);
        w9[j] = static_cast<const T *>(a.W) + co * 9 * CIN;
        w1[j] = static_cast<const T *>(a.W2) + co * CIN;
    }

    if (DRN_DBG(a) & 2) {
    } else if (SECOND == 3) {
        // start from what the 3x3 pass left in out (all loads of the lane issued together), add the 1x1 convolution
        const int gq = lane >> 4;
        const T *out = static_cast<const T *>(a.out);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int grow = mt < nm ? rowg[(mt0 + mt) * 16 + cq] : -1;
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (grow >= 0) load4<T>(out + (size_t)grow * COUT + (NTW * ng + j) * 16 + 4 * gq, v);
                acc[j][mt] = f32x4{v[0], v[1], v[2], v[3]};
            }
        }
        conv_accum<T, CIN, 1, NTW, MT>(acc, tileA, w1, hidx, nm, img.rowe, lane);
    } else {
        conv_accum<T, CIN, 9, NTW, MT>(acc, tileA, w9, hidx, nm, img.rowe, lane);
    }
    if (!(DRN_DBG(a) & 4))
    conv_epilogue<T, COUT, NTW, MT, EPI>(acc, static_cast<T *>(a.out), EPI ? a.stats + (size_t)blockIdx.x * NSTAT * COUT : nullptr, red,
                                         rowg, mt0, nm, ng, lane, wave, tid, a);
    if (SECOND == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[j][mt] = zero;
        conv_accum<T, CIN, 1, NTW, MT>(acc, tileA, w1, hidx, nm, img.rowe, lane);
        conv_epilogue<T, COUT, NTW, MT, EPI>(acc, static_cast<T *>(a.out2), a.stats2 + (size_t)blockIdx.x * 2 * COUT, red, rowg, mt0,
                                             nm, ng, lane, wave, tid, a);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// first convolution (1 -> 32 channels): VALU, x tile in LDS.  y0 [R,32] + statistics partials [blocks][2][32]
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void drn_conv0_kernel(const float *x, const float *w0, T *y0, float *stats, const Geom g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, W2 = g.tw + 2;
    float *xs = reinterpret_cast<float *>(smem);            // [F][HPt]
    float *red = xs + g.F * g.HPt;                          // [16][2][32]
    int *rowg = reinterpret_cast<int *>(red + 16 * 2 * 32), *rowc = rowg + g.RT;
    build_row_tables(g, blockIdx.x, g.RT, rowg, rowc, tid);
    for (int i = tid; i < g.F * g.HPt; i += NT) {
        const int src = cell_source(g, blockIdx.x, i);
        xs[i] = src >= 0 ? x[src] : 0.f;
    }
    __syncthreads();
    const int co = tid & 31, rs = tid >> 5;
    float w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = w0[co * 9 + t];
    float s = 0.f, ss = 0.f;
    for (int r = rs; r < g.RT; r += NT / 32) {
        const int grow = rowg[r];
        if (grow < 0) continue;
        const int h = rowc[r];
        float v = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) v += xs[h + (t / 3 - 1) * W2 + (t % 3 - 1)] * w[t];
        y0[(size_t)grow * 32 + co] = from_f32<T>(v);
        s += v; ss += v * v;
    }
    red[(rs * 2 + 0) * 32 + co] = s; red[(rs * 2 + 1) * 32 + co] = ss;
    __syncthreads();
    if (tid < 64) {
        float v = 0.f;
        for (int k = 0; k < NT / 32; ++k) v += red[(k * 2 + (tid >> 5)) * 32 + (tid & 31)];
        stats[(size_t)blockIdx.x * 64 + tid] = v;
    }
}

// weight gradient of the first convolution: dW0[co][tap] = sum_r dy0[r][co] * x[pixel(r) + tap]
// thread = (row set rs, 8-channel chunk cv): one 16-byte load of g and of y per row (a lane per channel made 2-byte loads:
// 357 us for 0.3 GB), all of a group's rows in flight before the first use; [9 taps][8 channels] partial sums in registers
template <typename T>
__global__ __launch_bounds__(NT) void drn_wgrad0_kernel(const float *x, const TileSrc d, float *slab, const Geom g, int ngroups) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int V = vec_el<T>(), CV = 32 / V, NRS = NT / CV, U = 4;
    const int tid = threadIdx.x, W2 = g.tw + 2, lane = tid & 63, wave = tid >> 6;
    float *xs = reinterpret_cast<float *>(smem);            // [F][HPt]
    float *red = xs + g.F * g.HPt;                          // [8 waves][288]
    int *rowg = reinterpret_cast<int *>(red + 16 * 288), *rowc = rowg + g.RT;
    const int c = (tid % CV) * V, rs = tid / CV;
    float k[V], c0[V], c1[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { k[e] = d.ca[c + e]; c0[e] = d.ca[CSTR + c + e]; c1[e] = d.ca[2 * CSTR + c + e]; }
    const T *gsrc = static_cast<const T *>(d.p0) + c, *ysrc = static_cast<const T *>(d.p1) + c;
    float acc[9][V];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[t][e] = 0.f;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        __syncthreads();
        build_row_tables(g, grp, g.RT, rowg, rowc, tid);
        for (int i = tid; i < g.F * g.HPt; i += NT) {
            const int src = cell_source(g, grp, i);
            xs[i] = src >= 0 ? x[src] : 0.f;
        }
        __syncthreads();
        for (int base = rs; base < g.RT; base += U * NRS) {
            int grow[U], h[U];
            uint4 rg[U], ry[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = base + u * NRS, rc = r < g.RT ? r : base;
                grow[u] = r < g.RT ? rowg[rc] : -1; h[u] = rowc[rc];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t row = grow[u] > 0 ? grow[u] : 0;
                rg[u] = *reinterpret_cast<const uint4 *>(gsrc + row * 32); ry[u] = *reinterpret_cast<const uint4 *>(ysrc + row * 32);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float gv[V], yv[V], xv[9];
                unpack16<T>(rg[u], gv); unpack16<T>(ry[u], yv);
#pragma unroll
                for (int t = 0; t < 9; ++t) xv[t] = grow[u] >= 0 ? xs[h[u] + (t / 3 - 1) * W2 + (t % 3 - 1)] : 0.f;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float dy = k[e] * gv[e] + c0[e] + c1[e] * yv[e];
#pragma unroll
                    for (int t = 0; t < 9; ++t) acc[t][e] += dy * xv[t];
                }
            }
        }
    }
    // lanes of a wave that share a channel chunk (lane % CV) -> lane < CV, then the 8 waves through LDS
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float v = acc[t][e];
#pragma unroll
            for (int m = CV; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
            acc[t][e] = v;
        }
    __syncthreads();
    if (lane < CV) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < V; ++e) red[wave * 288 + (c + e) * 9 + t] = acc[t][e];
    }
    __syncthreads();
    if (tid < 288) {
        float v = 0.f;
        for (int kk = 0; kk < NT / 64; ++kk) v += red[kk * 288 + tid];
        slab[(size_t)blockIdx.x * 288 + tid] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm statistics: partial [nb][2][C] -> forward table [mean | rstd | scale | shift] + running-statistics update
// ---------------------------------------------------------------------------------------------------------------
template <typename PT>
__global__ __launch_bounds__(NT) void drn_bn_finalize_kernel(const PT *part, int nb, int C, double count, const float *gamma,
                                                             const float *beta, float *rmean, float *rvar, float momentum,
                                                             float eps, float *coef, const double *count_dev) {
    __shared__ double red[2][4][128];
    if (count_dev) count = *count_dev;
    const int c = threadIdx.x & 127, sl = threadIdx.x >> 7;
    double s = 0.0, q = 0.0;
    if (c < C)
        for (int b = sl; b < nb; b += 4) { s += part[((size_t)b * 2 + 0) * C + c]; q += part[((size_t)b * 2 + 1) * C + c]; }
    red[0][sl][c] = s; red[1][sl][c] = q;
    __syncthreads();
    if (sl == 0 && c < C) {
        s = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        q = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        const double mean = s / count;
        double var = q / count - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float scale = gamma[c] * rstd;
        coef[c] = (float)mean; coef[CSTR + c] = rstd; coef[2 * CSTR + c] = scale; coef[3 * CSTR + c] = beta[c] - (float)mean * scale;
        if (rmean) {
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * count / (count > 1.0 ? count - 1.0 : 1.0));
        }
    }
}

// inference: forward table from the running statistics
__global__ void drn_bn_running_kernel(int C, const float *gamma, const float *beta, const float *rmean, const float *rvar, float eps,
                                      float *coef) {
    const int c = threadIdx.x;
    if (c >= C) return;
    const float rstd = 1.f / sqrtf(rvar[c] + eps), scale = gamma[c] * rstd;
    coef[c] = rmean[c]; coef[CSTR + c] = rstd; coef[2 * CSTR + c] = scale; coef[3 * CSTR + c] = beta[c] - rmean[c] * scale;
}

// partial [nb][W] -> [nb2][W]  (out[b2] = sum of the parts b = b2, b2 + nb2, ...), used when a kernel left many partials
__global__ void drn_part_reduce_kernel(const float *part, int nb, int W, float *out, int nb2) {
    const int w = blockIdx.y * blockDim.x + threadIdx.x, b2 = blockIdx.x;
    if (w >= W) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;          // four independent chains: the loads of a round are in flight together
    int b = b2;
    for (; b + 3 * nb2 < nb; b += 4 * nb2) {
        const float v0 = part[(size_t)b * W + w], v1 = part[(size_t)(b + nb2) * W + w];
        const float v2 = part[(size_t)(b + 2 * nb2) * W + w], v3 = part[(size_t)(b + 3 * nb2) * W + w];
        s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; b < nb; b += nb2) s0 += part[(size_t)b * W + w];
    out[(size_t)b2 * W + w] = (float)((s0 + s1) + (s2 + s3));
}

// synchronised BatchNorm: partial [nb][W] -> one fp64 row [W] (the all-reduced operand) and optionally a second copy
__global__ void drn_sync_reduce_kernel(const float *part, int nb, int W, double *out, double *copy) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += part[(size_t)b * W + w];
    out[w] = s;
    if (copy) copy[w] = s;
}

// ---------------------------------------------------------------------------------------------------------------
// global average pooling of o2 = relu(BN(y22) + BN(y2s)):  pooled [N,128] fp32
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(128) void drn_pool_kernel(const T *ya, const T *yb, const float *ca, const float *cb, float *pooled, int PP) {
    // one frame per workgroup; thread = (pixel set ps, 8-channel chunk): 16-byte loads, 4 pixels in flight per thread
    constexpr int V = vec_el<T>(), CV = 128 / V, NPS = 128 / CV, U = 4;
    __shared__ float red[NPS][128];
    const int c = (threadIdx.x % CV) * V, ps = threadIdx.x / CV;
    const size_t base = (size_t)blockIdx.x * PP * 128 + c;
    float sa[V], ha[V], sb[V], hb[V], acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        sa[e] = ca[2 * CSTR + c + e]; ha[e] = ca[3 * CSTR + c + e]; sb[e] = cb[2 * CSTR + c + e]; hb[e] = cb[3 * CSTR + c + e];
        acc[e] = 0.f;
    }
    for (int p0 = ps; p0 < PP; p0 += U * NPS) {
        uint4 ra[U], rb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = p0 + u * NPS < PP ? p0 + u * NPS : p0;
            ra[u] = *reinterpret_cast<const uint4 *>(ya + base + (size_t)p * 128); rb[u] = *reinterpret_cast<const uint4 *>(yb + base + (size_t)p * 128);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float a[V], b[V];
            unpack16<T>(ra[u], a); unpack16<T>(rb[u], b);
            const bool live = p0 + u * NPS < PP;
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += live ? fmaxf(sa[e] * a[e] + ha[e] + sb[e] * b[e] + hb[e], 0.f) : 0.f;
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) red[ps][c + e] = acc[e];
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NPS; ++k) s += red[k][threadIdx.x];
    pooled[(size_t)blockIdx.x * 128 + threadIdx.x] = s / (float)PP;
}

// ---------------------------------------------------------------------------------------------------------------
// backward, step 1 of every BatchNorm: g = upstream * [activation > 0]  (written), and per-channel partial sums
//      s1 = sum g,  sa = sum g*y,  sb = sum g*y'      -> part [blocks][3][C]
// TOP: upstream = dpooled[frame][c] / (P*P) (gradient of the average pooling) instead of a [R,C] tensor
// ---------------------------------------------------------------------------------------------------------------
struct MaskArgs {
    const void *up; const void *y; const void *y2; const float *ca; const float *cb;
    void *g; float *part; int64_t R; int PP; int rows_per_block;
};

template <typename T, int C, bool DUAL, bool TOP>
__global__ __launch_bounds__(NT) void drn_mask_reduce_kernel(const MaskArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int V = vec_el<T>(), CV = C / V, NRS = NT / CV;
    float *red = reinterpret_cast<float *>(smem);             // [NRS][3][C]
    const int tid = threadIdx.x, cv = tid % CV, rs = tid / CV, c = cv * V;
    float sA[V], hA[V], sB[V], hB[V], s1[V], sa[V], sb[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        sA[e] = a.ca[2 * CSTR + c + e]; hA[e] = a.ca[3 * CSTR + c + e];
        sB[e] = DUAL ? a.cb[2 * CSTR + c + e] : 0.f; hB[e] = DUAL ? a.cb[3 * CSTR + c + e] : 0.f;
        s1[e] = sa[e] = sb[e] = 0.f;
    }
    const int64_t rbeg = (int64_t)blockIdx.x * a.rows_per_block;
    const int64_t rend = rbeg + a.rows_per_block < a.R ? rbeg + a.rows_per_block : a.R;
    const T *y = static_cast<const T *>(a.y), *y2 = static_cast<const T *>(a.y2);
    T *g = static_cast<T *>(a.g);
    const float inv = 1.f / (float)a.PP;
    for (int64_t r = rbeg + rs; r < rend; r += NRS) {
        float yv[V], y2v[V], up[V], gv[V];
        load16(y + r * C + c, yv);
        if (DUAL) load16(y2 + r * C + c, y2v);
        if (TOP) {
            const float *dp = static_cast<const float *>(a.up) + (r / a.PP) * C + c;
#pragma unroll
            for (int e = 0; e < V; ++e) up[e] = dp[e] * inv;
        } else {
            load16(static_cast<const T *>(a.up) + r * C + c, up);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float act = sA[e] * yv[e] + hA[e];
            if (DUAL) act += sB[e] * y2v[e] + hB[e];
            gv[e] = act > 0.f ? up[e] : 0.f;
            s1[e] += gv[e]; sa[e] += gv[e] * yv[e];
            if (DUAL) sb[e] += gv[e] * y2v[e];
        }
        store16(g + r * C + c, gv);
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
        red[(rs * 3 + 0) * C + c + e] = s1[e]; red[(rs * 3 + 1) * C + c + e] = sa[e]; red[(rs * 3 + 2) * C + c + e] = sb[e];
    }
    __syncthreads();
    for (int t = tid; t < 3 * C; t += NT) {
        float v = 0.f;
        for (int k = 0; k < NRS; ++k) v += red[k * 3 * C + t];
        a.part[(size_t)blockIdx.x * 3 * C + t] = v;
    }
}

// partial [nb][3][C] -> d gamma, d beta and the DY table [k | c0 | c1]:  dy = k*g + c0 + c1*y
// `local` (synchronised BatchNorm): this rank's own sums [3][C]; d gamma / d beta come from those while the table uses
// the sums and the count of the whole job
template <typename PT>
__global__ __launch_bounds__(NT) void drn_bn_bwd_finalize_kernel(const PT *part, int nb, int C, int which, double count,
                                                                 const float *gamma, const float *fcoef, float *bcoef,
                                                                 float *dgamma, float *dbeta, const PT *local,
                                                                 const double *count_dev) {
    __shared__ double red[2][4][128];
    if (count_dev) count = *count_dev;
    const int c = threadIdx.x & 127, sl = threadIdx.x >> 7;
    double s = 0.0, q = 0.0;
    if (c < C)
        for (int b = sl; b < nb; b += 4) { s += part[((size_t)b * 3 + 0) * C + c]; q += part[((size_t)b * 3 + which) * C + c]; }
    red[0][sl][c] = s; red[1][sl][c] = q;
    __syncthreads();
    if (sl == 0 && c < C) {
        s = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        q = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        const double mean = fcoef[c], rstd = fcoef[CSTR + c];
        const double dg = rstd * (q - mean * s);
        const double k = (double)gamma[c] * rstd;
        const double c1 = -k * rstd * dg / count;
        bcoef[c] = (float)k; bcoef[CSTR + c] = (float)(-k * s / count - c1 * mean); bcoef[2 * CSTR + c] = (float)c1;
        if (local) {
            const double sl_ = (double)local[c], ql_ = (double)local[(size_t)which * C + c];
            dgamma[c] = (float)(rstd * (ql_ - mean * sl_)); dbeta[c] = (float)sl_;
        } else {
            dgamma[c] = (float)dg; dbeta[c] = (float)s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient: dW[co][tap][ci] = sum_r dy[r][co] * a[pixel(r) + tap][ci]      (contraction over pixels)
// grid (G, TAPS/TPP): a workgroup walks frame groups blockIdx.x, +G, ... holding TPP taps (one kernel row) of the whole
// [COUT x CIN] block in accumulators; 8 waves = 4 (c_out) x 2 (c_in)
// ---------------------------------------------------------------------------------------------------------------
struct WgradArgs {
    Geom g;
    int ngroups;
    TileSrc A;       // input activation of the convolution (ACT1 / ACT2)
    TileSrc D;       // dy (DY)
    float *slab;     // [gridDim.x][COUT][TAPS][CIN]
};


// grid (G, CSPLIT): a workgroup walks frame groups blockIdx.x, +G, ... holding ALL taps of its c_out window
// (COUT / CSPLIT channels) x all c_in in accumulators, so every activation is read once per c_out window.
template <typename T, int CIN, int COUT, int TAPS, int PROA, int CSPLIT>
__global__ __launch_bounds__(NT) void drn_wgrad_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // 8 waves = WGM (c_out) x WGN (c_in).  The dy fragments do not depend on the tap, so tall wave tiles (more c_out tiles
    // per wave) need fewer LDS reads per MFMA
    constexpr int COB = COUT / CSPLIT;
    constexpr int WGM = CIN >= 64 ? 2 : 4, WGN = 8 / WGM;
    constexpr int WM = COB / 16 / WGM, WN = CIN / 16 / WGN;
    static_assert(WM >= 1 && WN >= 1 && TAPS * WM * WN <= 36, "wave tile / accumulator budget");
    // LDS: the activation image in the Img layout and dy rows 32 B apart modulo 256: the transposing reads take 8
    // consecutive rows x 32 B per half wave, conflict-free when consecutive rows (cells) sit 8 banks apart
    // (before: 16 B of padding, SQ_LDS_BANK_CONFLICT = 41-44 % of SQ_LDS_IDX_ACTIVE)
    constexpr int CSA = Img<T, CIN>::CS, CSD = COB + Img<T, COB>::PADE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, cq = lane & 15, q = cq >> 2, p = cq & 3;
    const int wm = wave % WGM, wn = wave / WGM;
    const int co_off = blockIdx.y * COB;
    const Geom &gm = a.g;
    const int RP = (gm.RT + 31) / 32 * 32;
    const Img<T, CIN> img(gm.th, gm.tw);
    T *tileA = reinterpret_cast<T *>(smem);                     // Img image: zero-haloed input activation
    T *tileD = tileA + gm.F * img.slote;                        // [RP][CSD]     dy window, plain row order, dead rows zero
    int *rowg = reinterpret_cast<int *>(tileD + RP * CSD), *hmap = rowg + RP, *cellsrc = hmap + RP;   // [RP], [RP], [F*HPt]
    int *cellpos = cellsrc + gm.F * gm.HPt;                     // [F*HPt]
    f32x4 acc[TAPS][WM][WN];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[t][i][j] = zero;
    for (int grp = blockIdx.x; grp < a.ngroups; grp += gridDim.x) {
        __syncthreads();
        build_row_tables(gm, grp, RP, rowg, hmap, tid);
        build_cell_tables(gm, grp, cellsrc, cellpos, CSA, img.rowe, img.slote, tid);
        __syncthreads();
        for (int r = tid; r < RP; r += NT) hmap[r] = cellpos[hmap[r]];          // cell index -> element offset in the image
        fill_image<T, CIN, PROA, (PROA == PRO_ACT2 ? 2 : 4)>(tileA, a.A, gm, cellsrc, cellpos, tid);
        fill_rows<T, COB, COUT, PRO_DY>(tileD, a.D, rowg, RP, co_off, tid);
        __syncthreads();
        const int ksteps = RP / 32;
#pragma unroll 1
        for (int ks = 0; ks < ksteps; ++ks) {
            const int r0 = ks * 32;
            if constexpr (sizeof(T) == 2) {
                const int rlo = r0 + 4 * g + q, rhi = rlo + 16;
                bf16x8 af[WM];
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    const int co0 = (wm * WM + i) * 16 + 4 * p;
                    af[i] = tr_pair(reinterpret_cast<const bf16 *>(tileD) + rlo * CSD + co0, reinterpret_cast<const bf16 *>(tileD) + rhi * CSD + co0);
                }
                const int hl = hmap[rlo], hh = hmap[rhi];
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const int off = TAPS == 9 ? ((t / 3 - 1) * img.rowe + (t % 3 - 1) * CSA) : 0;
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        const int ci0 = (wn * WN + j) * 16 + 4 * p;
                        const bf16x8 bfr = tr_pair(reinterpret_cast<const bf16 *>(tileA) + hl + off + ci0,
                                                   reinterpret_cast<const bf16 *>(tileA) + hh + off + ci0);
#pragma unroll
                        for (int i = 0; i < WM; ++i) acc[t][i][j] = Mma<bf16>::mma(af[i], bfr, acc[t][i][j]);
                    }
                }
            } else {
#pragma unroll 2
                for (int ss = 0; ss < 8; ++ss) {
                    const int r = r0 + 4 * ss + g;
                    float af[WM];
#pragma unroll
                    for (int i = 0; i < WM; ++i) af[i] = reinterpret_cast<const float *>(tileD)[r * CSD + (wm * WM + i) * 16 + cq];
                    const int h = hmap[r];
#pragma unroll
                    for (int t = 0; t < TAPS; ++t) {
                        const int off = TAPS == 9 ? ((t / 3 - 1) * img.rowe + (t % 3 - 1) * CSA) : 0;
#pragma unroll
                        for (int j = 0; j < WN; ++j) {
                            const float bfr = reinterpret_cast<const float *>(tileA)[h + off + (wn * WN + j) * 16 + cq];
#pragma unroll
                            for (int i = 0; i < WM; ++i) acc[t][i][j] = Mma<float>::mma(af[i], bfr, acc[t][i][j]);
                        }
                    }
                }
            }
        }
    }
    float *slab = a.slab + (size_t)blockIdx.x * COUT * TAPS * CIN;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co_off + (wm * WM + i) * 16 + 4 * g + r, ci = (wn * WN + j) * 16 + cq;
                    slab[((size_t)co * TAPS + t) * CIN + ci] = acc[t][i][j][r];
                }
    }
}

// slab [nparts][COUT][TAPS][CIN] -> dW in the reference layout [COUT][CIN][TAPS] (Conv2d weight, taps = ky*3+kx)
__global__ void drn_wgrad_reduce_kernel(const float *slab, int nparts, int COUT, int TAPS, int CIN, float *dW) {
    const int n = COUT * TAPS * CIN, idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;          // four chains (fixed order: still bitwise reproducible), 4 loads in flight
    int k = 0;
    for (; k + 3 < nparts; k += 4) {
        const float v0 = slab[(size_t)k * n + idx], v1 = slab[(size_t)(k + 1) * n + idx];
        const float v2 = slab[(size_t)(k + 2) * n + idx], v3 = slab[(size_t)(k + 3) * n + idx];
        s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; k < nparts; ++k) s0 += slab[(size_t)k * n + idx];
    const float s = (s0 + s1) + (s2 + s3);
    const int ci = idx % CIN, tap = (idx / CIN) % TAPS, co = idx / (CIN * TAPS);
    dW[((size_t)co * CIN + ci) * TAPS + tap] = s;
}

// reference layout [COUT][CIN][TAPS] fp32 -> forward pack [COUT][TAPS][CIN] and data-gradient pack [CIN][TAPS-1-tap][COUT]
template <typename T>
__global__ void drn_pack_kernel(const float *W, int COUT, int CIN, int TAPS, T *Wf, T *Wd) {
    const int n = COUT * CIN * TAPS, idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int tap = idx % TAPS, ci = (idx / TAPS) % CIN, co = idx / (TAPS * CIN);
    const T v = from_f32<T>(W[idx]);
    Wf[((size_t)co * TAPS + tap) * CIN + ci] = v;
    Wd[((size_t)ci * TAPS + (TAPS - 1 - tap)) * COUT + co] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct DrnWs {
    size_t y[7], wf[7], wd[7], fcoef, bcoef, pooled, dpooled, part, part2, X[3], slab, lin, total;
    int parts_cap;
};

constexpr int MASK_ROWS = 512, WG_GROUPS = 256, PART_MAX = 128;

// tile side: whole frame when it fits (bf16: <= 13, fp32: <= 9 -- two 128-channel images must fit 160 KB), else the
// frame is cut into the fewest equal tiles per dimension
void choose_tile(int dtype, int P, int *t, int *nt) {
    const int tsmax = dtype == MIVIT_BF16 ? 13 : 9;
    *nt = ceil_div(P, tsmax);
    *t = ceil_div(P, *nt);
}
Geom make_geom(int dtype, int N, int P, int F) {
    Geom g{};
    int t, nt;
    choose_tile(dtype, P, &t, &nt);
    g.N = N; g.P = P; g.PP = P * P; g.th = t; g.tw = t; g.ntx = nt; g.tpf = nt * nt; g.units = N * g.tpf;
    g.F = F; g.HPt = (t + 2) * (t + 2); g.RT = F * t * t;
    return g;
}

// slots per workgroup: as many as fit mm * 16 output pixels and need(F) <= cap bytes of LDS
template <typename NeedFn>
int slots_fit(int t, NeedFn need, int mm = MAXM, size_t cap = (size_t)160 * 1024) {
    int F = (mm * 16) / (t * t);
    while (F >= 1 && need(F) > cap) --F;
    return F;
}
// bytes of one Img<T, C> image of F slots (same arithmetic as the device struct)
size_t conv_image_bytes(int dtype, int t, int F, int C) {
    const int es = (int)dtype_size(dtype);
    if (es != 2) return (size_t)F * (t + 2) * (t + 2) * (C + 4) * es;
    const int csb = (C + 16) * es, rowb = (t + 2) * csb + (256 - (2 * csb) % 256) % 256;
    const int spb = (((t * t * csb - (t + 2) * rowb) % 256) + 256) % 256;
    return (size_t)F * ((size_t)(t + 2) * rowb + spb);
}
size_t conv_lds(int dtype, int t, int F, int CIN, int CIN2, int mm = MAXM) {
    return conv_image_bytes(dtype, t, F, CIN) + (CIN2 ? conv_image_bytes(dtype, t, F, CIN2) : 0) + 8 * 3 * 32 * 4 + 2 * mm * 16 * 4 +
           (size_t)F * (t + 2) * (t + 2) * 4 +
           (size_t)F * (t + 2) * (t + 2) * 4;
}
size_t wgrad_lds(int dtype, int t, int F, int CIN, int COUT /* c_out window held in LDS */) {
    const int RP = (F * t * t + 31) / 32 * 32;
    return conv_image_bytes(dtype, t, F, CIN) + (size_t)RP * (COUT * dtype_size(dtype) + (dtype_size(dtype) == 2 ? 32 : 16)) +
           (size_t)RP * 8 + (size_t)2 * F * (t + 2) * (t + 2) * 4;
}
int conv0_slots(int t) { return std::max(1, 512 / (t * t)); }

bool drn_train_supported(int dtype, int P) {
    if (P < 1 || P > 4096) return false;
    int t, nt;
    choose_tile(dtype, P, &t, &nt);
    return slots_fit(t, [&](int F) { return conv_lds(dtype, t, F, 128, 0); }) >= 1 &&
           slots_fit(t, [&](int F) { return wgrad_lds(dtype, t, F, 128, 64); }) >= 1;
}

DrnWs make_ws(int dtype, int N, int P, int E) {
    DrnWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes, 256); return o; };
    const size_t R = (size_t)N * P * P, es = dtype_size(dtype);
    for (int i = 0; i < 7; ++i) w.y[i] = take(R * DRN_CO[i] * es);
    for (int i = 0; i < 7; ++i) { w.wf[i] = take((size_t)DRN_CO[i] * DRN_CI[i] * DRN_TAPS[i] * es); w.wd[i] = take((size_t)DRN_CO[i] * DRN_CI[i] * DRN_TAPS[i] * es); }
    w.fcoef = take(7 * 4 * CSTR * 4);
    w.bcoef = take(7 * 3 * CSTR * 4);
    w.pooled = take((size_t)N * 128 * 4);
    w.dpooled = take((size_t)N * 128 * 4);
    int t, nt;
    choose_tile(dtype, P, &t, &nt);
    w.parts_cap = std::max(N * nt * nt, (int)((R + MASK_ROWS - 1) / MASK_ROWS)) + 8;
    w.part = take((size_t)w.parts_cap * 3 * 128 * 4 * 2);
    w.part2 = take((size_t)PART_MAX * 3 * 128 * 4);
    for (int i = 0; i < 3; ++i) w.X[i] = take(R * 128 * es);
    w.slab = take((size_t)WG_GROUPS / 2 * 128 * 9 * 128 * 4);   // 128 parts of the 128x128x9 gradient = 256 of a 64x128x9 one
    w.lin = take(linear_wgrad_ws_bytes(N, E, 128));
    w.total = off;
    return w;
}

inline void *at(void *ws, size_t off) { return static_cast<char *>(ws) + off; }

template <typename K>
int set_lds(K kernel, size_t bytes) {
    MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}
#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

struct Ctx {
    int dtype, N, P, E;
    float eps, momentum;
    void *ws; DrnWs w; hipStream_t s;
    const mivit_deepresnet_params *prm;
    bool use_running = false;      // inference: BatchNorm with the running statistics
    // synchronised BatchNorm (staged entry points): `stats` = [2][3][128] fp64 exchanged by the caller between stages
    double *stats = nullptr; int stage = -1; const double *gcount = nullptr;
    bool sync() const { return stats != nullptr; }
    bool in(int st) const { return !stats || st == stage; }
    double *stat(int k) const { return stats + (size_t)k * 3 * 128; }
    float *fco(int i) const { return static_cast<float *>(at(ws, w.fcoef)) + (size_t)i * 4 * CSTR; }
    float *bco(int i) const { return static_cast<float *>(at(ws, w.bcoef)) + (size_t)i * 3 * CSTR; }
    void *y(int i) const { return at(ws, w.y[i]); }
    float *part() const { return static_cast<float *>(at(ws, w.part)); }
    double count() const { return (double)N * P * P; }
};

// many partials -> at most PART_MAX (deterministic), returns the pointer / count the finalize kernels should read
int squeeze_parts(const Ctx &c, const float *&part, int &nb, int W) {
    if (nb <= PART_MAX) return 0;
    float *out = static_cast<float *>(at(c.ws, c.w.part2));
    hipLaunchKernelGGL(drn_part_reduce_kernel, dim3(PART_MAX, ceil_div(W, 128)), dim3(128), 0, c.s, part, nb, W, out, PART_MAX);
    MIVIT_LAUNCH_CHECK();
    part = out; nb = PART_MAX;
    return 0;
}

int bn_finalize(const Ctx &c, int i, const float *part, int nb) {
    const int C = DRN_CO[i];
    const mivit_conv_bn &b = c.prm->conv[i];
    if (c.use_running) {
        hipLaunchKernelGGL(drn_bn_running_kernel, dim3(1), dim3(128), 0, c.s, C, b.gamma, b.beta, b.running_mean, b.running_var,
                           c.eps, c.fco(i));
        MIVIT_LAUNCH_CHECK();
        return 0;
    }
    RC(squeeze_parts(c, part, nb, 2 * C));
    hipLaunchKernelGGL(drn_bn_finalize_kernel<float>, dim3(1), dim3(NT), 0, c.s, part, nb, C, c.count(), b.gamma, b.beta, b.running_mean,
                       b.running_var, c.momentum, c.eps, c.fco(i), (const double *)nullptr);
    MIVIT_LAUNCH_CHECK();
    return 0;
}

// after a convolution: either finish BatchNorm i from this rank's partials, or (synchronised) leave the local sums in
// stats slot k for the caller to all-reduce
int bn_produce(const Ctx &c, int i, int k, const float *part, int nb) {
    if (!c.sync()) return bn_finalize(c, i, part, nb);
    const int W = 2 * DRN_CO[i];
    hipLaunchKernelGGL(drn_sync_reduce_kernel, dim3(ceil_div(W, 128)), dim3(128), 0, c.s, part, nb, W, c.stat(k), (double *)nullptr);
    MIVIT_LAUNCH_CHECK();
    return 0;
}
// before BatchNorm i is applied: (synchronised) finish it from the all-reduced sums in slot k and the job-wide count
int bn_consume(const Ctx &c, int i, int k) {
    if (!c.sync()) return 0;
    const mivit_conv_bn &b = c.prm->conv[i];
    hipLaunchKernelGGL(drn_bn_finalize_kernel<double>, dim3(1), dim3(NT), 0, c.s, c.stat(k), 1, DRN_CO[i], 0.0, b.gamma, b.beta,
                       b.running_mean, b.running_var, c.momentum, c.eps, c.fco(i), c.gcount);
    MIVIT_LAUNCH_CHECK();
    return 0;
}

template <typename T, int CIN, int COUT, int SECOND, int PRO, int EPI, int MM, int NTW>
int run_conv_mm(const Ctx &c, const ConvArgs &proto, int *nblocks, int t, int F) {
    ConvArgs a = proto;
    // a launcher variant whose tile does not fit its LDS / row-tile budget must report it, never divide by it
    MIVIT_CHECK(F >= 1 && F * t * t <= MM * 16, "deepresnet conv %d->%d: no frame slot fits (tile %d, %d row tiles, F=%d)", CIN, COUT, t, MM, F);
#ifdef MIVIT_DRN_PHASE_TIMING   // phase-timing build only (results are wrong with a switch set): never defined in the shipped library
    { static const int dbg = [] { const char *e = getenv("MIVIT_DRN_DBG"); return e ? atoi(e) : 0; }(); a.dbg = dbg; }
#endif
    a.g = make_geom(c.dtype, c.N, c.P, F);
    const size_t lds = conv_lds(c.dtype, t, F, CIN, 0, MM);
    auto kern = drn_conv_kernel<T, CIN, COUT, SECOND, PRO, EPI, MM, NTW>;
    RC(set_lds(kern, lds));
    const int blocks = ceil_div(a.g.units, F);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds, c.s, a);
    MIVIT_LAUNCH_CHECK();
    if (nblocks) *nblocks = blocks;
    return 0;
}

template <typename T, int CIN, int COUT, int SECOND, int PRO, int EPI>
int run_conv(const Ctx &c, const ConvArgs &proto, int *nblocks) {
    // one workgroup per CU with as many frames as fit: half-size groups, two co-resident per CU (MM = MAXM / 2, <= 80 KB,
    // <= 128 VGPRs), measured 4-5 % slower at 9 x 9 (bf16) and 20 % slower in fp32 -- the weights are re-streamed per group
    int t, nt;
    choose_tile(c.dtype, c.P, &t, &nt);
    const int F = slots_fit(t, [&](int f) { return conv_lds(c.dtype, t, f, CIN, 0); });
    // measured (B=1024 bf16, us under the profiler, NTW 2 -> 1): 128->128 992 -> 910, 128->64 890 -> 660, 64->64 501 -> 390,
    // 64->32 405 -> 319; whole step 13.44 -> 12.75 ms with NTW = 1 everywhere (MIVIT_DRN_NTW=2 selects the old wave tile)
    static const int ntw = [] { const char *e = getenv("MIVIT_DRN_NTW"); return e ? atoi(e) : 1; }();
    if (ntw == 2) return run_conv_mm<T, CIN, COUT, SECOND, PRO, EPI, MAXM, 2>(c, proto, nblocks, t, F);
    if constexpr (SECOND == 3 && sizeof(T) == 2) {
        // the 1x1 skip pass is memory-bound and its weights are 16 KB: half-size groups, two workgroups per CU (<= 80 KB,
        // <= 128 VGPRs), so one group's fill runs under the other's epilogue (bf16 step 12.34 -> 12.04 ms; fp32: no gain)
        static const int half3 = [] { const char *e = getenv("MIVIT_DRN_SKIP_HALF"); return e ? atoi(e) : 1; }();
        constexpr int MMH = MAXM / 2;
        const int Fh = slots_fit(t, [&](int f) { return conv_lds(c.dtype, t, f, CIN, 0, MMH); }, MMH, (size_t)80 * 1024);
        if (half3 && Fh >= 1) return run_conv_mm<T, CIN, COUT, SECOND, PRO, EPI, MMH, 1>(c, proto, nblocks, t, Fh);
    }
    return run_conv_mm<T, CIN, COUT, SECOND, PRO, EPI, MAXM, 1>(c, proto, nblocks, t, F);
}

template <typename T>
int pack_weights(const Ctx &c) {
    for (int i = 1; i < 7; ++i) {
        const int n = DRN_CO[i] * DRN_CI[i] * DRN_TAPS[i];
        hipLaunchKernelGGL(drn_pack_kernel<T>, dim3(ceil_div(n, 256)), dim3(256), 0, c.s, c.prm->conv[i].weight, DRN_CO[i], DRN_CI[i],
                           DRN_TAPS[i], static_cast<T *>(at(c.ws, c.w.wf[i])), static_cast<T *>(at(c.ws, c.w.wd[i])));
        MIVIT_LAUNCH_CHECK();
    }
    return 0;
}

template <typename T>
int forward_t(const Ctx &c, const float *x, float *tokens) {
    // stages (synchronised BatchNorm runs one per call, the caller all-reduces c.stats in between; otherwise all six run)
    float *part = c.part(), *part_b = part + (size_t)c.w.parts_cap * 3 * 128;
    int nb = 0;
    ConvArgs a{};
    if (c.in(0)) {   // conv0
        RC(pack_weights<T>(c));
        int t, nt;
        choose_tile(c.dtype, c.P, &t, &nt);
        const Geom g = make_geom(c.dtype, c.N, c.P, conv0_slots(t));
        nb = ceil_div(g.units, g.F);
        const size_t lds = (size_t)g.F * g.HPt * 4 + 16 * 2 * 32 * 4 + (size_t)2 * g.RT * 4;
        hipLaunchKernelGGL(drn_conv0_kernel<T>, dim3(nb), dim3(NT), lds, c.s, x, c.prm->conv[0].weight, static_cast<T *>(c.y(0)), part, g);
        MIVIT_LAUNCH_CHECK();
        RC(bn_produce(c, 0, 0, part, nb));
    }
    if (c.in(1)) {   // block 1: conv1 (32->64) + skip (1x1) from a0 = relu(BN0(y0))
        RC(bn_consume(c, 0, 0));
        a = ConvArgs{};
        a.A = TileSrc{c.y(0), nullptr, c.fco(0), nullptr};
        a.W = at(c.ws, c.w.wf[1]); a.W2 = at(c.ws, c.w.wf[3]); a.out = c.y(1); a.out2 = c.y(3); a.stats = part; a.stats2 = part_b;
        RC((run_conv<T, 32, 64, 1, PRO_ACT1, 1>(c, a, &nb)));
        RC(bn_produce(c, 1, 0, part, nb)); RC(bn_produce(c, 3, 1, part_b, nb));
    }
    if (c.in(2)) {   // block 1: conv2 (64->64) from relu(BN1(y11))
        RC(bn_consume(c, 1, 0)); RC(bn_consume(c, 3, 1));
        a = ConvArgs{};
        a.A = TileSrc{c.y(1), nullptr, c.fco(1), nullptr};
        a.W = at(c.ws, c.w.wf[2]); a.out = c.y(2); a.stats = part;
        RC((run_conv<T, 64, 64, 0, PRO_ACT1, 1>(c, a, &nb)));
        RC(bn_produce(c, 2, 0, part, nb));
    }
    if (c.in(3)) {   // block 2: conv1 (64->128) + skip from o1 = relu(BN2(y12) + BN3(y1s))
        RC(bn_consume(c, 2, 0));
        a = ConvArgs{};
        a.A = TileSrc{c.y(2), c.y(3), c.fco(2), c.fco(3)};
        a.W = at(c.ws, c.w.wf[4]); a.W2 = at(c.ws, c.w.wf[6]); a.out = c.y(4); a.out2 = c.y(6); a.stats = part; a.stats2 = part_b;
        RC((run_conv<T, 64, 128, 1, PRO_ACT2, 1>(c, a, &nb)));
        RC(bn_produce(c, 4, 0, part, nb)); RC(bn_produce(c, 6, 1, part_b, nb));
    }
    if (c.in(4)) {   // block 2: conv2 (128->128)
        RC(bn_consume(c, 4, 0)); RC(bn_consume(c, 6, 1));
        a = ConvArgs{};
        a.A = TileSrc{c.y(4), nullptr, c.fco(4), nullptr};
        a.W = at(c.ws, c.w.wf[5]); a.out = c.y(5); a.stats = part;
        RC((run_conv<T, 128, 128, 0, PRO_ACT1, 1>(c, a, &nb)));
        RC(bn_produce(c, 5, 0, part, nb));
    }
    if (!c.in(5)) return 0;
    RC(bn_consume(c, 5, 0));
    // pooling + fc
    float *pooled = static_cast<float *>(at(c.ws, c.w.pooled));
    hipLaunchKernelGGL(drn_pool_kernel<T>, dim3(c.N), dim3(128), 0, c.s, static_cast<const T *>(c.y(5)), static_cast<const T *>(c.y(6)),
                       c.fco(5), c.fco(6), pooled, c.P * c.P);
    MIVIT_LAUNCH_CHECK();
    LinearFwdArgs l{};
    l.dtype = MIVIT_F32; l.x = pooled; l.x_is_f32 = 1; l.ldx = 128; l.W = c.prm->fc_weight; l.bias = c.prm->fc_bias;
    l.M = c.N; l.N = c.E; l.K = 128; l.act = MIVIT_ACT_NONE; l.y = tokens; l.ldy = c.E; l.y_is_f32 = 1;
    return launch_linear_fwd(l, c.s);
}

template <typename T, int C, bool DUAL, bool TOP>
int run_mask(const Ctx &c, const void *up, int ya, int yb, void *g, int *nblocks) {
    constexpr int V = vec_el<T>(), NRS = NT / (C / V);
    MaskArgs m{up, c.y(ya), DUAL ? c.y(yb) : nullptr, c.fco(ya), DUAL ? c.fco(yb) : nullptr, g, c.part(), (int64_t)c.N * c.P * c.P,
               c.P * c.P, MASK_ROWS};
    const int blocks = (int)((m.R + MASK_ROWS - 1) / MASK_ROWS);
    const size_t lds = (size_t)NRS * 3 * C * 4;
    auto kern = drn_mask_reduce_kernel<T, C, DUAL, TOP>;
    RC(set_lds(kern, lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds, c.s, m);
    MIVIT_LAUNCH_CHECK();
    *nblocks = blocks;
    return 0;
}

int bn_bwd_finalize(const Ctx &c, int i, int which, int nb, const mivit_deepresnet_grads *gr) {
    const int C = DRN_CO[i];
    const float *part = c.part();
    RC(squeeze_parts(c, part, nb, 3 * C));
    hipLaunchKernelGGL(drn_bn_bwd_finalize_kernel<float>, dim3(1), dim3(NT), 0, c.s, part, nb, C, which, c.count(), c.prm->conv[i].gamma,
                       c.fco(i), c.bco(i), gr->conv[i].gamma, gr->conv[i].beta, (const float *)nullptr, (const double *)nullptr);
    MIVIT_LAUNCH_CHECK();
    return 0;
}
// after a mask pass over C channels: (synchronised) local sums [3][C] -> stats slot 0 (all-reduced) and slot 1 (kept local)
int bwd_produce(const Ctx &c, int C, int nb) {
    if (!c.sync()) return 0;
    hipLaunchKernelGGL(drn_sync_reduce_kernel, dim3(ceil_div(3 * C, 128)), dim3(128), 0, c.s, c.part(), nb, 3 * C, c.stat(0), c.stat(1));
    MIVIT_LAUNCH_CHECK();
    return 0;
}
// gradient table + d gamma / d beta of BatchNorm i: from this rank's partials, or (synchronised) from the exchanged sums
int bwd_consume(const Ctx &c, int i, int which, int nb, const mivit_deepresnet_grads *gr) {
    if (!c.sync()) return bn_bwd_finalize(c, i, which, nb, gr);
    hipLaunchKernelGGL(drn_bn_bwd_finalize_kernel<double>, dim3(1), dim3(NT), 0, c.s, c.stat(0), 1, DRN_CO[i], which, 0.0,
                       c.prm->conv[i].gamma, c.fco(i), c.bco(i), gr->conv[i].gamma, gr->conv[i].beta, (const double *)c.stat(1), c.gcount);
    MIVIT_LAUNCH_CHECK();
    return 0;
}

template <typename T, int CIN, int COUT, int TAPS, int PROA>
int run_wgrad(const Ctx &c, const TileSrc &A, const TileSrc &D, float *dW) {
    // all taps in one pass; the 128 x 128 convolution splits c_out in two windows to fit the accumulators
    constexpr int CSPLIT = (TAPS * (COUT / 16) * (CIN / 16) / 8 > 36) ? 2 : 1;
    WgradArgs a{};
    int t, nt;
    choose_tile(c.dtype, c.P, &t, &nt);
    const int F = slots_fit(t, [&](int f) { return wgrad_lds(c.dtype, t, f, CIN, COUT / CSPLIT); });
    MIVIT_CHECK(F >= 1, "deepresnet wgrad %d->%d: no frame slot fits LDS (tile %d)", CIN, COUT, t);
    a.g = make_geom(c.dtype, c.N, c.P, F);
    a.ngroups = ceil_div(a.g.units, F);
    a.A = A; a.D = D; a.slab = static_cast<float *>(at(c.ws, c.w.slab));
    const int G = std::min(a.ngroups, WG_GROUPS / CSPLIT);        // one resident workgroup per CU
    const size_t lds = wgrad_lds(c.dtype, t, F, CIN, COUT / CSPLIT);
    auto kern = drn_wgrad_kernel<T, CIN, COUT, TAPS, PROA, CSPLIT>;
    RC(set_lds(kern, lds));
    hipLaunchKernelGGL(kern, dim3(G, CSPLIT), dim3(NT), lds, c.s, a);
    MIVIT_LAUNCH_CHECK();
    const int n = COUT * TAPS * CIN;
    hipLaunchKernelGGL(drn_wgrad_reduce_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, c.s, a.slab, G, COUT, TAPS, CIN, dW);
    MIVIT_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int backward_t(const Ctx &c, const float *x, const float *dtokens, const mivit_deepresnet_grads *gr) {
    // stages as in forward_t: each ends with a mask pass whose sums the next stage's BatchNorm gradient table needs
    float *pooled = static_cast<float *>(at(c.ws, c.w.pooled)), *dpooled = static_cast<float *>(at(c.ws, c.w.dpooled));
    void *X1 = at(c.ws, c.w.X[0]), *X2 = at(c.ws, c.w.X[1]), *X3 = at(c.ws, c.w.X[2]);
    int nb = 0;
    ConvArgs a{};
    const TileSrc dy22{X1, c.y(5), c.bco(5), nullptr}, dy2s{X1, c.y(6), c.bco(6), nullptr};
    const TileSrc a21{c.y(4), nullptr, c.fco(4), nullptr}, o1{c.y(2), c.y(3), c.fco(2), c.fco(3)};
    const TileSrc dy21{X2, c.y(4), c.bco(4), nullptr};
    const TileSrc dy12{X3, c.y(2), c.bco(2), nullptr}, dy1s{X3, c.y(3), c.bco(3), nullptr};
    const TileSrc a11{c.y(1), nullptr, c.fco(1), nullptr}, a0{c.y(0), nullptr, c.fco(0), nullptr};
    const TileSrc dy11{X1, c.y(1), c.bco(1), nullptr};
    if (c.in(0)) {
        // fc: dpooled = dtokens W ; dW = dtokens^T pooled ; db
        LinearDgradArgs d{};
        d.dtype = MIVIT_F32; d.dy = dtokens; d.dy_is_f32 = 1; d.lddy = c.E; d.W = c.prm->fc_weight; d.M = c.N; d.N = c.E; d.K = 128;
        d.act = MIVIT_ACT_NONE; d.dx = dpooled; d.lddx = 128; d.dx_is_f32 = 1;
        RC(launch_linear_dgrad(d, c.s));
        LinearWgradArgs wg{};
        wg.dtype = MIVIT_F32; wg.dy = dtokens; wg.dy_is_f32 = 1; wg.lddy = c.E; wg.x = pooled; wg.x_is_f32 = 1; wg.ldx = 128;
        wg.M = c.N; wg.N = c.E; wg.K = 128; wg.dW = gr->fc_weight; wg.db = gr->fc_bias; wg.ws = at(c.ws, c.w.lin);
        wg.ws_bytes = linear_wgrad_ws_bytes(c.N, c.E, 128);
        RC(launch_linear_wgrad(wg, c.s));
        // ---- top: g2 = pool-gradient * [o2 > 0] -> X1 ; BatchNorm 5 (conv2) and 6 (skip) of block 2
        RC((run_mask<T, 128, true, true>(c, dpooled, 5, 6, X1, &nb)));
        RC(bwd_produce(c, 128, nb));
    }
    if (c.in(1)) {
        RC(bwd_consume(c, 5, 1, nb, gr)); RC(bwd_consume(c, 6, 2, nb, gr));
        RC((run_wgrad<T, 128, 128, 9, PRO_ACT1>(c, a21, dy22, gr->conv[5].weight)));
        RC((run_wgrad<T, 64, 128, 1, PRO_ACT2>(c, o1, dy2s, gr->conv[6].weight)));
        a = ConvArgs{};
        // d a21, masked by a21's ReLU in the epilogue -> g21 in X2, with the sums for BatchNorm 4 (block 2 conv1)
        a.A = dy22; a.W = at(c.ws, c.w.wd[5]); a.out = X2;
        a.my = c.y(4); a.mca = c.fco(4); a.stats = c.part();
        RC((run_conv<T, 128, 128, 0, PRO_DY, 2>(c, a, &nb)));
        RC(bwd_produce(c, 128, nb));
    }
    if (c.in(2)) {
        RC(bwd_consume(c, 4, 1, nb, gr));
        RC((run_wgrad<T, 64, 128, 9, PRO_ACT2>(c, o1, dy21, gr->conv[4].weight)));
        a = ConvArgs{};
        // d o1 -> X3 [R,64]: the 3x3 pass, then the skip's 1x1 pass adds to it
        a.A = dy21; a.W = at(c.ws, c.w.wd[4]); a.out = X3;
        RC((run_conv<T, 128, 64, 0, PRO_DY, 0>(c, a, nullptr)));
        // ... whose epilogue applies block 1's output ReLU -> g1 in X3, with the sums for BatchNorm 2 (conv2) and 3 (skip)
        a = ConvArgs{};
        a.A = dy2s; a.W2 = at(c.ws, c.w.wd[6]); a.out = X3;
        a.my = c.y(2); a.my2 = c.y(3); a.mca = c.fco(2); a.mcb = c.fco(3); a.stats = c.part();
        RC((run_conv<T, 128, 64, 3, PRO_DY, 3>(c, a, &nb)));
        RC(bwd_produce(c, 64, nb));
    }
    if (c.in(3)) {
        RC(bwd_consume(c, 2, 1, nb, gr)); RC(bwd_consume(c, 3, 2, nb, gr));
        RC((run_wgrad<T, 64, 64, 9, PRO_ACT1>(c, a11, dy12, gr->conv[2].weight)));
        RC((run_wgrad<T, 32, 64, 1, PRO_ACT1>(c, a0, dy1s, gr->conv[3].weight)));
        a = ConvArgs{};
        a.A = dy12; a.W = at(c.ws, c.w.wd[2]); a.out = X1;                                   // d a11, masked -> g11 in X1 [R,64]
        a.my = c.y(1); a.mca = c.fco(1); a.stats = c.part();
        RC((run_conv<T, 64, 64, 0, PRO_DY, 2>(c, a, &nb)));
        RC(bwd_produce(c, 64, nb));
    }
    if (c.in(4)) {
        RC(bwd_consume(c, 1, 1, nb, gr));
        RC((run_wgrad<T, 32, 64, 9, PRO_ACT1>(c, a0, dy11, gr->conv[1].weight)));
        a = ConvArgs{};
        a.A = dy11; a.W = at(c.ws, c.w.wd[1]); a.out = X2;                                     // d a0 -> X2 [R,32]
        RC((run_conv<T, 64, 32, 0, PRO_DY, 0>(c, a, nullptr)));
        a = ConvArgs{};
        a.A = dy1s; a.W2 = at(c.ws, c.w.wd[3]); a.out = X2;                                    // + skip term, masked -> g0
        a.my = c.y(0); a.mca = c.fco(0); a.stats = c.part();
        RC((run_conv<T, 64, 32, 3, PRO_DY, 2>(c, a, &nb)));
        RC(bwd_produce(c, 32, nb));
    }
    if (c.in(5)) {   // first convolution's weight gradient
        RC(bwd_consume(c, 0, 1, nb, gr));
        int t, nt;
        choose_tile(c.dtype, c.P, &t, &nt);
        const Geom g = make_geom(c.dtype, c.N, c.P, conv0_slots(t));
        const int ngroups = ceil_div(g.units, g.F), G = std::min(ngroups, 256);
        const size_t lds = (size_t)g.F * g.HPt * 4 + 16 * 288 * 4 + (size_t)2 * g.RT * 4;
        float *slab = static_cast<float *>(at(c.ws, c.w.slab));
        const TileSrc dy0{X2, c.y(0), c.bco(0), nullptr};
        hipLaunchKernelGGL(drn_wgrad0_kernel<T>, dim3(G), dim3(NT), lds, c.s, x, dy0, slab, g, ngroups);
        MIVIT_LAUNCH_CHECK();
        RC(launch_slab_reduce(slab, G, 288, gr->conv[0].weight, 0, c.s));
    }
    return 0;
}

constexpr int64_t GRAPH_MAX_ROWS = 1 << 17;      // below this many pixels a step is launch-bound: replay it as a hipGraph

std::vector<uint64_t> params_key(const mivit_deepresnet_params *p) {
    std::vector<uint64_t> k;
    for (int i = 0; i < 7; ++i)
        for (const float *q : {p->conv[i].weight, p->conv[i].gamma, p->conv[i].beta, (const float *)p->conv[i].running_mean,
                               (const float *)p->conv[i].running_var})
            k.push_back((uint64_t)q);
    k.push_back((uint64_t)p->fc_weight); k.push_back((uint64_t)p->fc_bias);
    return k;
}

int check_params(const mivit_deepresnet_params *p) {
    MIVIT_CHECK(p && p->fc_weight && p->fc_bias, "deepresnet: null fc parameters");
    for (int i = 0; i < 7; ++i)
        MIVIT_CHECK(p->conv[i].weight && p->conv[i].gamma && p->conv[i].beta, "deepresnet: null parameter in conv/bn %d", i);
    return 0;
}

}  // namespace

extern "C" int mivit_deepresnet_train_supported(int dtype, int patch_size) { return drn_train_supported(dtype, patch_size) ? 1 : 0; }

extern "C" size_t mivit_deepresnet_train_workspace_bytes(int dtype, int N, int P, int E) {
    if (N <= 0 || E <= 0 || !drn_train_supported(dtype, P)) return 0;
    return make_ws(dtype, N, P, E).total;
}

// introspection for tests / diagnostics: byte offsets of the workspace regions
// [0..6] raw convolution outputs y0..y6, [7] forward tables, [8] gradient tables, [9] pooled, [10] dpooled, [11] partials,
// [12..14] gradient buffers X1..X3, [15] total
extern "C" int mivit_deepresnet_train_workspace_layout(int dtype, int N, int P, int E, size_t *offsets) {
    MIVIT_CHECK(offsets, "deepresnet_train_workspace_layout: null pointer");
    if (N <= 0 || E <= 0 || !drn_train_supported(dtype, P)) { mivit_set_error("deepresnet_train_workspace_layout: unsupported shape"); return 3; }
    const DrnWs w = make_ws(dtype, N, P, E);
    for (int i = 0; i < 7; ++i) offsets[i] = w.y[i];
    offsets[7] = w.fcoef; offsets[8] = w.bcoef; offsets[9] = w.pooled; offsets[10] = w.dpooled; offsets[11] = w.part;
    for (int i = 0; i < 3; ++i) offsets[12 + i] = w.X[i];
    offsets[15] = w.total;
    return 0;
}

extern "C" int mivit_deepresnet_train_fwd(int dtype, const mivit_deepresnet_params *params, const float *x, int N, int P, int E,
                                          float momentum, float eps, float *tokens, void *workspace, size_t workspace_bytes,
                                          void *stream) {
    MIVIT_CHECK(dtype == MIVIT_F32 || dtype == MIVIT_BF16, "bad dtype %d", dtype);
    RC(check_params(params));
    MIVIT_CHECK(x && tokens && workspace, "deepresnet_train_fwd: null pointer");
    MIVIT_CHECK(N > 0 && E > 0, "deepresnet_train_fwd: empty problem");
    if (!drn_train_supported(dtype, P)) { mivit_set_error("deepresnet_train_fwd: unsupported frame side %d", P); return 3; }
    Ctx c{dtype, N, P, E, eps, momentum, workspace, make_ws(dtype, N, P, E), static_cast<hipStream_t>(stream), params};
    MIVIT_CHECK(workspace_bytes >= c.w.total, "deepresnet_train_fwd: workspace too small (%zu < %zu)", workspace_bytes, c.w.total);
    prof_set_tag(MIVIT_PROF_OP);
    auto body = [&](hipStream_t cs) {
        Ctx cc = c; cc.s = cs;
        return dtype == MIVIT_F32 ? forward_t<float>(cc, x, tokens) : forward_t<bf16>(cc, x, tokens);
    };
    if ((int64_t)N * P * P > GRAPH_MAX_ROWS) return body(c.s);
    std::vector<uint64_t> key = params_key(params);
    for (uint64_t v : {(uint64_t)3, (uint64_t)dtype, (uint64_t)x, (uint64_t)N, (uint64_t)P, (uint64_t)E, (uint64_t)tokens, (uint64_t)workspace,
                       (uint64_t)__builtin_bit_cast(uint32_t, momentum), (uint64_t)__builtin_bit_cast(uint32_t, eps)})
        key.push_back(v);
    return graph_run(key.data(), (int)key.size(), c.s, body);
}

extern "C" int mivit_deepresnet_train_bwd(int dtype, const mivit_deepresnet_params *params, const float *x, const float *dtokens,
                                          int N, int P, int E, float eps, const mivit_deepresnet_grads *grads, void *workspace,
                                          size_t workspace_bytes, void *stream) {
    MIVIT_CHECK(dtype == MIVIT_F32 || dtype == MIVIT_BF16, "bad dtype %d", dtype);
    RC(check_params(params));
    MIVIT_CHECK(x && dtokens && workspace && grads && grads->fc_weight && grads->fc_bias, "deepresnet_train_bwd: null pointer");
    for (int i = 0; i < 7; ++i)
        MIVIT_CHECK(grads->conv[i].weight && grads->conv[i].gamma && grads->conv[i].beta, "deepresnet_train_bwd: null gradient %d", i);
    if (!drn_train_supported(dtype, P)) { mivit_set_error("deepresnet_train_bwd: unsupported frame side %d", P); return 3; }
    Ctx c{dtype, N, P, E, eps, 0.f, workspace, make_ws(dtype, N, P, E), static_cast<hipStream_t>(stream), params};
    MIVIT_CHECK(workspace_bytes >= c.w.total, "deepresnet_train_bwd: workspace too small (%zu < %zu)", workspace_bytes, c.w.total);
    prof_set_tag(MIVIT_PROF_OP);
    auto body = [&](hipStream_t cs) {
        Ctx cc = c; cc.s = cs;
        return dtype == MIVIT_F32 ? backward_t<float>(cc, x, dtokens, grads) : backward_t<bf16>(cc, x, dtokens, grads);
    };
    if ((int64_t)N * P * P > GRAPH_MAX_ROWS) return body(c.s);
    std::vector<uint64_t> key = params_key(params);
    for (int i = 0; i < 7; ++i)
        for (const float *p : {grads->conv[i].weight, grads->conv[i].gamma, grads->conv[i].beta}) key.push_back((uint64_t)p);
    for (uint64_t v : {(uint64_t)4, (uint64_t)dtype, (uint64_t)x, (uint64_t)dtokens, (uint64_t)N, (uint64_t)P, (uint64_t)E, (uint64_t)workspace,
                       (uint64_t)grads->fc_weight, (uint64_t)grads->fc_bias, (uint64_t)__builtin_bit_cast(uint32_t, eps)})
        key.push_back(v);
    return graph_run(key.data(), (int)key.size(), c.s, body);
}

// Synchronised BatchNorm under data parallelism: the same forward / backward cut into six stages each.  A stage leaves
// this rank's BatchNorm sums in `stats` ([2][3][128] fp64, device); the caller all-reduces them (forward: the whole
// buffer; backward: the first [3][128] only -- the second keeps the local sums d gamma / d beta are made from) and calls
// the next stage, which finishes the statistics with *global_count (device fp64) = frames of the whole job x P^2.
extern "C" int mivit_deepresnet_train_fwd_stage(int dtype, const mivit_deepresnet_params *params, const float *x, int N, int P, int E,
                                                float momentum, float eps, float *tokens, void *workspace, size_t workspace_bytes,
                                                int stage, const double *global_count, double *stats, void *stream) {
    MIVIT_CHECK(dtype == MIVIT_F32 || dtype == MIVIT_BF16, "bad dtype %d", dtype);
    RC(check_params(params));
    MIVIT_CHECK(x && tokens && workspace && stats && global_count, "deepresnet_train_fwd_stage: null pointer");
    MIVIT_CHECK(N > 0 && E > 0, "deepresnet_train_fwd_stage: empty problem");
    MIVIT_CHECK(stage >= 0 && stage < MIVIT_DEEPRESNET_STAGES, "deepresnet_train_fwd_stage: bad stage %d", stage);
    if (!drn_train_supported(dtype, P)) { mivit_set_error("deepresnet_train_fwd_stage: unsupported frame side %d", P); return 3; }
    Ctx c{dtype, N, P, E, eps, momentum, workspace, make_ws(dtype, N, P, E), static_cast<hipStream_t>(stream), params};
    MIVIT_CHECK(workspace_bytes >= c.w.total, "deepresnet_train_fwd_stage: workspace too small (%zu < %zu)", workspace_bytes, c.w.total);
    c.stats = stats; c.stage = stage; c.gcount = global_count;
    prof_set_tag(MIVIT_PROF_OP);
    return dtype == MIVIT_F32 ? forward_t<float>(c, x, tokens) : forward_t<bf16>(c, x, tokens);
}

extern "C" int mivit_deepresnet_train_bwd_stage(int dtype, const mivit_deepresnet_params *params, const float *x, const float *dtokens,
                                                int N, int P, int E, float eps, const mivit_deepresnet_grads *grads, void *workspace,
                                                size_t workspace_bytes, int stage, const double *global_count, double *stats, void *stream) {
    MIVIT_CHECK(dtype == MIVIT_F32 || dtype == MIVIT_BF16, "bad dtype %d", dtype);
    RC(check_params(params));
    MIVIT_CHECK(x && dtokens && workspace && stats && global_count && grads && grads->fc_weight && grads->fc_bias, "deepresnet_train_bwd_stage: null pointer");
    for (int i = 0; i < 7; ++i)
        MIVIT_CHECK(grads->conv[i].weight && grads->conv[i].gamma && grads->conv[i].beta, "deepresnet_train_bwd_stage: null gradient %d", i);
    MIVIT_CHECK(stage >= 0 && stage < MIVIT_DEEPRESNET_STAGES, "deepresnet_train_bwd_stage: bad stage %d", stage);
    if (!drn_train_supported(dtype, P)) { mivit_set_error("deepresnet_train_bwd_stage: unsupported frame side %d", P); return 3; }
    Ctx c{dtype, N, P, E, eps, 0.f, workspace, make_ws(dtype, N, P, E), static_cast<hipStream_t>(stream), params};
    MIVIT_CHECK(workspace_bytes >= c.w.total, "deepresnet_train_bwd_stage: workspace too small (%zu < %zu)", workspace_bytes, c.w.total);
    c.stats = stats; c.stage = stage; c.gcount = global_count;
    prof_set_tag(MIVIT_PROF_OP);
    return dtype == MIVIT_F32 ? backward_t<float>(c, x, dtokens, grads) : backward_t<bf16>(c, x, dtokens, grads);
}

extern "C" int mivit_deepresnet_infer(int dtype, const mivit_deepresnet_params *params, const float *x, int N, int P, int E,
                                      float eps, float *tokens, void *workspace, size_t workspace_bytes, void *stream) {
    MIVIT_CHECK(dtype == MIVIT_F32 || dtype == MIVIT_BF16, "bad dtype %d", dtype);
    RC(check_params(params));
    for (int i = 0; i < 7; ++i)
        MIVIT_CHECK(params->conv[i].running_mean && params->conv[i].running_var, "deepresnet_infer: running statistics required");
    MIVIT_CHECK(x && tokens && workspace, "deepresnet_infer: null pointer");
    MIVIT_CHECK(N > 0 && E > 0, "deepresnet_infer: empty problem");
    if (!drn_train_supported(dtype, P)) { mivit_set_error("deepresnet_infer: unsupported frame side %d", P); return 3; }
    Ctx c{dtype, N, P, E, eps, 0.f, workspace, make_ws(dtype, N, P, E), static_cast<hipStream_t>(stream), params};
    c.use_running = true;
    MIVIT_CHECK(workspace_bytes >= c.w.total, "deepresnet_infer: workspace too small (%zu < %zu)", workspace_bytes, c.w.total);
    prof_set_tag(MIVIT_PROF_OP);
    return dtype == MIVIT_F32 ? forward_t<float>(c, x, tokens) : forward_t<bf16>(c, x, tokens);
}
