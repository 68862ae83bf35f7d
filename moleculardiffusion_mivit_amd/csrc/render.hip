// Noise-free rendering of single-particle frames from trajectories: the GPU counterpart of the
// reference's Python triple loop trajectories_to_video -> trajectory_to_video -> gaussian_2d + block_reduce
// (helpers/helpersGeneration.py:128-319: per frame, nPosPerFrame Gaussian spots evaluated on a grid `up` times finer than
// the camera, each rescaled so its PEAK on that grid equals its intensity (:305-308), summed, then mean-pooled up x up).
//
// A 2-D Gaussian on the fine grid, its peak and the up x up mean are all separable, so one workgroup per (sequence, frame,
// PSF width) builds the two 1-D pooled, peak-normalised profiles of every sub-position in LDS
//     px[p][i] = mean_{k < up} exp(-(axis[i up + k] - x_p)^2 / 2 s^2) / max_g exp(-(axis[g] - x_p)^2 / 2 s^2)
// (the maximum is the value at the grid point nearest to x_p, clamped into the grid) and then writes
//     frame[y][x] = sum_p a_p py[p][y] px[p][x].
// HBM traffic: the trajectory segment in, P x P floats out -- the 25x finer grid never exists in memory.
#include "common.h"

namespace {

struct RenderArgs {
    const float *traj;        // [N, T, 2] positions in camera pixels (x, y)
    const float *sigmas;      // [nsig] Gaussian sigma on the fine grid
    const float *amp;         // [N, F, npos] spot intensities
    float *out;               // [N, nsig, F, P, P]
    int N, T, npos, nsig, P, up, center;
};

__global__ __launch_bounds__(256) void render_frames_kernel(const RenderArgs a) {
    extern __shared__ float sm[];
    const int f = blockIdx.x, n = blockIdx.y, si = blockIdx.z;
    const int F = a.T / a.npos, P = a.P, up = a.up, G = P * up, npos = a.npos;
    float *px = sm, *py = sm + npos * P, *amp = py + npos * P, *cen = amp + npos;      // cen[2]
    const float s = a.sigmas[si], inv2s2 = 1.f / (2.f * s * s);
    const int limit = (G - 1) / 2;
    const float step = G > 1 ? 2.f * (float)limit / (float)(G - 1) : 0.f;
    const float *seg = a.traj + ((int64_t)n * a.T + (int64_t)f * npos) * 2;
    if (threadIdx.x < 2) {
        float m = 0.f;
        if (a.center) {
            for (int p = 0; p < npos; ++p) m += seg[2 * p + threadIdx.x];
            m /= (float)npos;
        }
        cen[threadIdx.x] = m;
    }
    for (int p = threadIdx.x; p < npos; p += blockDim.x) amp[p] = a.amp[((int64_t)n * F + f) * npos + p];
    __syncthreads();
    // 1-D profiles: task = (axis, sub-position, camera pixel)
    for (int t = threadIdx.x; t < 2 * npos * P; t += blockDim.x) {
        const int ax = t / (npos * P), r = t - ax * npos * P, p = r / P, i = r - p * P;
        const float c = (seg[2 * p + ax] - cen[ax]) * (float)up;
        // peak of the spot on the fine grid: the grid point nearest to c (clamped)
        float gi = step > 0.f ? rintf((c + (float)limit) / step) : 0.f;
        gi = fminf(fmaxf(gi, 0.f), (float)(G - 1));
        const float dpk = (-(float)limit + gi * step) - c;
        // spot / spot_max in one exponential: exp(-(d^2 - dpk^2) / 2 s^2).  The reference divides two float64 Gaussians, so a
        // spot that has left the frame (spot_max ~ 1e-50) still comes out at full intensity on the border; the difference of
        // squares keeps that behaviour in fp32, where the two factors alone would underflow to 0 / 0.
        float acc = 0.f;
        for (int k = 0; k < up; ++k) {
            const float d = (-(float)limit + (float)(i * up + k) * step) - c;
            acc += __expf(-(d * d - dpk * dpk) * inv2s2);
        }
        (ax == 0 ? px : py)[p * P + i] = acc / (float)up;
    }
    __syncthreads();
    float *dst = a.out + (((int64_t)n * a.nsig + si) * F + f) * P * P;
    for (int t = threadIdx.x; t < P * P; t += blockDim.x) {
        const int y = t / P, x = t - y * P;
        float v = 0.f;
        for (int p = 0; p < npos; ++p) v += amp[p] * py[p * P + y] * px[p * P + x];
        dst[t] = v;
    }
}

}  // namespace

// frames[N, nsig, F, P, P] (fp32) from trajectories [N, T, 2] (camera pixels), F = T / npos frames of npos sub-positions each
extern "C" int mivit_render_frames(const float *traj_px, int N, int T, int npos, const float *sigmas, int nsig, int P, int up,
                                   const float *amp, int center, float *out, void *stream) {
    MIVIT_CHECK(traj_px && sigmas && amp && out, "render_frames: null pointer");
    MIVIT_CHECK(N > 0 && T > 0 && npos > 0 && nsig > 0 && P > 0 && up > 0, "render_frames: empty problem");
    MIVIT_CHECK(T % npos == 0, "T is not divisble by posPerFrame");
    MIVIT_CHECK(N <= 65535 && nsig <= 65535, "render_frames: more than 65535 sequences / PSF widths per launch");
    const size_t lds = ((size_t)2 * npos * P + npos + 2) * sizeof(float);
    MIVIT_CHECK(lds <= 64 * 1024, "render_frames: %d sub-positions x %d pixels do not fit LDS", npos, P);
    RenderArgs a{traj_px, sigmas, amp, out, N, T, npos, nsig, P, up, center};
    prof_set_tag(MIVIT_PROF_OP);
    hipLaunchKernelGGL(render_frames_kernel, dim3(T / npos, N, nsig), dim3(256), lds, static_cast<hipStream_t>(stream), a);
    MIVIT_LAUNCH_CHECK();
    return 0;
}
