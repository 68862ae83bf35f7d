// Device primitives shared by the streaming kernels (embed, rowstream, wavestream, gemm_dma, wgrad_dma, wgrad_small,
// deepresnet_train, attention_fast): LDS-DMA issue, counted vmcnt waits, raw barrier, bf16 MFMA, transposed LDS reads.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;

// global -> LDS, 16 bytes per lane: the LDS destination is wave-uniform base + lane * 16, the source address is per lane.
// The instruction is WRITTEN OUT rather than issued through __builtin_amdgcn_global_load_lds: the compiler's waitcnt pass cannot
// tell which LDS bytes a pending LDS-DMA will write, so before the next ds_write -- and before LDS reads it cannot disambiguate,
// e.g. the transposing reads of a weight image staged long ago -- it inserts s_waitcnt vmcnt(0): a wait for the prefetch just
// issued (the ring never overlaps anything) and for the acks of every global store before it.  Inline asm is invisible to
// that pass; unknown extra VM operations can only make its own counted waits stricter, never looser (they retire in order).
// EVERY consumer therefore orders its read-back with an explicit wait_vm<N>() (+ barrier() where other waves' shares matter).
__device__ __forceinline__ void dma16(const void *g, void *l) {
    typedef __attribute__((address_space(3))) void lptr_t;
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t *)l);
    // s_nop 0: an SALU write of M0 needs one wait state before an LDS-DMA reads it; the hazard recogniser pads the builtin form
    // but cannot see inside an asm string
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(base) : "memory", "m0");
}
__device__ __forceinline__ void dma16_opaque(const void *g, void *l) { dma16(g, l); }
// the same with the source as a wave-uniform 64-bit base (SGPR pair) + a 32-bit per-lane byte offset: no 64-bit per-lane
// pointer has to live in VGPRs across a loop (kernels at their register limit: mlp_block_bwd8)
__device__ __forceinline__ void dma16_sbase(const void *base_uniform, unsigned off_bytes, unsigned lds_addr_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off_bytes), "s"(base_uniform), "s"(lds_addr_uniform) : "memory", "m0");
}
__device__ __forceinline__ void dma4_sbase(const void *base_uniform, unsigned off_bytes, unsigned lds_addr_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(off_bytes), "s"(base_uniform), "s"(lds_addr_uniform) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr(const void *l) {          // 32-bit LDS byte address of a shared-memory pointer
    typedef __attribute__((address_space(3))) void lptr_t;
    return (unsigned)(size_t)(lptr_t *)l;
}
// the builtin form (the compiler tracks it).  Kept for the frame-embedding weight gradient, the one kernel that measured
// SLOWER with the written-out form (1.39 -> 1.53 ms: its loop only reads LDS, nothing was being over-waited, and the
// compiler's own placement of the M0 set-up and waits is the better schedule there).
__device__ __forceinline__ void dma16_tracked(const void *g, void *l) {
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    __builtin_amdgcn_global_load_lds((gptr_t *)g, (lptr_t *)l, 16, 0, 0);
}
// 4 bytes per lane (LDS destination = wave-uniform base + lane * 4): per-row scalars of a prefetched tile.  A register-returning
// load carried across the loop's back edge makes the compiler wait vmcnt(0) at its first use -- i.e. for the stores issued since.
__device__ __forceinline__ void dma4_opaque(const void *g, void *l) {
    typedef __attribute__((address_space(3))) void lptr_t;
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t *)l);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(base) : "memory", "m0");
}
// vector-memory operations (loads, LDS-DMA, stores) retire in order: N = how many of this wave's youngest may still be in flight.
// Every call site states the VM program order it relies on (prologue, steady state, tail); tests/test_wait_model.py replays
// those orders on the host and checks every count.  -DMIVIT_STRICT_WAITS (libmivit_hip_strict.so, csrc/build.py) turns every
// counted wait into vmcnt(0): tests/test_strict_waits_gpu.py requires bitwise-equal results from the two libraries.
template <int N>
__device__ __forceinline__ void wait_vm() {
#ifdef MIVIT_STRICT_WAITS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
// workgroup barrier without the compiler's blanket vmcnt(0): LDS traffic drained, DMA left in flight
__device__ __forceinline__ void barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
// ordering of one wave's own LDS writes before its cross-lane reads (wave-private scratch)
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(f16x8 a, f16x8 b, f32x4 c) {          // (the -DMIVIT_ELEM_F16 builds of the streaming kernels: elem.h)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// MFMA operand whose k index runs over ROWS of a natural [row][col] bf16 LDS image: each lane passes the address of
// (row base + q, column base + 4p) for the low and the high half of its 8 k slots (q = (lane & 15) >> 2, p = lane & 3)
// and receives column base + (lane & 15) of those rows (ds_read_b64_tr_b16)
__device__ __forceinline__ bf16x8 tr_pair(const bf16 *lo, const bf16 *hi) {
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(lo));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(hi));
    struct { s16x4 a, b; } pr = {a, b};
    return __builtin_bit_cast(bf16x8, pr);
}
__device__ __forceinline__ f16x8 tr_pair_f16(const f16 *lo, const f16 *hi) {
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(lo));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(hi));
    struct { s16x4 a, b; } pr = {a, b};
    return __builtin_bit_cast(f16x8, pr);
}
