// Element type of the streaming kernels, chosen PER TRANSLATION UNIT.
//
// rowstream.hip, wavestream.hip, wgrad_dma.hip, wgrad_small.hip, attention_fast.hip, embed.hip and the fused encoder-layer
// blocks fused_fwd.hip / fused_bwd.hip are written against the
// 16-bit storage type `bf16` (struct h16<0>), its vector form `bf16x8` and the compiler's `__bf16`.  csrc/build.py compiles
// each of them twice: as is, and with -DMIVIT_ELEM_F16, where the three names below stand for IEEE half (`f16`, `f16x8`,
// `_Float16`) and every external function of the unit carries the suffix _f16.  Everything type-specific in those files goes
// through overloads that exist for both types (to_f32 / from_f32 / load16 / store16 in common.h, mma / tr_pair in
// stream_prims.h) or through the macro and the two unpack helpers at the end; nothing else in them manipulates 16-bit patterns by hand.  The engine
// picks the bf16 or the _f16 set by the plan's dtype (engine.hip::stream_ops): BASELINE config 5's "fp16 with loss scaling"
// then runs on the same streaming kernels as bf16 instead of the general register-staged ones.
// Include AFTER common.h and stream_prims.h.
#pragma once
#ifdef MIVIT_ELEM_F16
#define MIVIT_ELEM_DTYPE MIVIT_F16
#define bf16 f16
#define bf16x8 f16x8
#define __bf16 _Float16
#define tr_pair tr_pair_f16
// externals of the six units
#define rowstream_supported rowstream_supported_f16
#define launch_rowstream launch_rowstream_f16
#define wavestream_supported wavestream_supported_f16
#define launch_wavestream launch_wavestream_f16
#define wgrad_dma_supported wgrad_dma_supported_f16
#define wgrad_dma_ws_bytes wgrad_dma_ws_bytes_f16
#define launch_wgrad_dma launch_wgrad_dma_f16
#define embed_small_fwd_supported embed_small_fwd_supported_f16
#define launch_embed_small_fwd launch_embed_small_fwd_f16
#define embed_small_wgrad_supported embed_small_wgrad_supported_f16
#define embed_small_wgrad_ws_bytes embed_small_wgrad_ws_bytes_f16
#define launch_embed_small_wgrad launch_embed_small_wgrad_f16
#define wgrad_small_supported wgrad_small_supported_f16
#define wgrad_small_ws_bytes wgrad_small_ws_bytes_f16
#define launch_wgrad_small launch_wgrad_small_f16
#define attention_fast_supported attention_fast_supported_f16
#define launch_attention_fwd_fast launch_attention_fwd_fast_f16
#define launch_attention_bwd_fast launch_attention_bwd_fast_f16
#define embed_dma_supported embed_dma_supported_f16
#define launch_embed_fwd_dma launch_embed_fwd_dma_f16
#define embed_wgrad_dma_ws_bytes embed_wgrad_dma_ws_bytes_f16
#define launch_embed_wgrad_dma launch_embed_wgrad_dma_f16
// the 16-deep MFMA of attention_fast.hip's backward (operands travel as 4 x 16-bit lanes)
#define ELEM_MFMA_16x16x16(a, b, c)                                                                                       \
    __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(__attribute__((ext_vector_type(4))) _Float16, a),            \
                                          __builtin_bit_cast(__attribute__((ext_vector_type(4))) _Float16, b), c, 0, 0, 0)
// the two 16-bit elements packed in one 32-bit word, widened to fp32 (the fused blocks unpack 8- and 16-byte loads by hand)
__device__ __forceinline__ float elem_lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
__device__ __forceinline__ float elem_hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
#else
#define MIVIT_ELEM_DTYPE MIVIT_BF16
#define ELEM_MFMA_16x16x16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0)
__device__ __forceinline__ float elem_lo(uint32_t w) { return __uint_as_float(w << 16); }            // bf16 = the high half of an fp32
__device__ __forceinline__ float elem_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
#endif

// The fused encoder-layer blocks (fused_fwd.hip / fused_bwd.hip) are in addition compiled per LAYER WIDTH: as is for
// E = 128 / F = 256 / head dim 32, and with -DMIVIT_WIDTH64 for the reference's shipped E = 64 / F = 128 / head dim 16
// (Experiments/Framerate/trainSettingsFramerate.py:42-47).  Their externals carry _w64 and / or _f16; the bf16 builds also
// export the operator-level C entry points (mivit_attn_block_fwd ..., include/mivit_hip.h), the width-64 one as ..._w64.
#if defined(MIVIT_WIDTH64) && defined(MIVIT_ELEM_F16)
#define MIVIT_FUSED_NAME(x) x##_w64_f16
#elif defined(MIVIT_WIDTH64)
#define MIVIT_FUSED_NAME(x) x##_w64
#elif defined(MIVIT_ELEM_F16)
#define MIVIT_FUSED_NAME(x) x##_f16
#endif
#ifdef MIVIT_FUSED_NAME
#define fused_layer_supported MIVIT_FUSED_NAME(fused_layer_supported)
#define launch_attn_block_fwd MIVIT_FUSED_NAME(launch_attn_block_fwd)
#define launch_mlp_block_fwd MIVIT_FUSED_NAME(launch_mlp_block_fwd)
#define mlp_block_bwd_ws_bytes MIVIT_FUSED_NAME(mlp_block_bwd_ws_bytes)
#define launch_mlp_block_bwd MIVIT_FUSED_NAME(launch_mlp_block_bwd)
#define attn_out_bwd_ws_bytes MIVIT_FUSED_NAME(attn_out_bwd_ws_bytes)
#define launch_attn_out_bwd MIVIT_FUSED_NAME(launch_attn_out_bwd)
#define qkv_bwd_ws_bytes MIVIT_FUSED_NAME(qkv_bwd_ws_bytes)
#define launch_qkv_bwd MIVIT_FUSED_NAME(launch_qkv_bwd)
#endif
#if defined(MIVIT_WIDTH64) && !defined(MIVIT_ELEM_F16)
#define mivit_fused_layer_supported mivit_fused_layer_supported_w64
#define mivit_mlp_block_fwd mivit_mlp_block_fwd_w64
#define mivit_attn_block_fwd mivit_attn_block_fwd_w64
#define mivit_mlp_block_bwd_workspace_bytes mivit_mlp_block_bwd_workspace_bytes_w64
#define mivit_mlp_block_bwd mivit_mlp_block_bwd_w64
#define mivit_attn_out_bwd_workspace_bytes mivit_attn_out_bwd_workspace_bytes_w64
#define mivit_attn_out_bwd mivit_attn_out_bwd_w64
#define mivit_qkv_bwd_workspace_bytes mivit_qkv_bwd_workspace_bytes_w64
#define mivit_qkv_bwd mivit_qkv_bwd_w64
#endif
