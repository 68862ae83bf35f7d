"""ctypes binding of libmivit_hip.so (the C-ABI declared in include/mivit_hip.h).

There is NO fallback: if the library is missing or does not load, importing this module raises, and every
product entry point that needs it fails loudly.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p, POINTER, Structure

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libmivit_hip.so")

ABI_VERSION = 1
F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LEAKY_RELU, ACT_GELU = 0, 1, 2, 3
EMBED_LINEAR, EMBED_CNN, EMBED_EXTERNAL = 0, 1, 2
FUSION_NONE, FUSION_EARLY, FUSION_LATE = 0, 1, 2


class MivitConfig(Structure):
    _fields_ = [(n, c_int) for n in (
        "abi_version", "dtype", "embedding", "patch_size", "embed_dim", "num_heads", "hidden_dim", "num_layers",
        "activation", "use_pos_encoding", "use_regression_token", "fusion", "global_feature_dim", "head_hidden",
        "output_dim")]


class ConvBn(Structure):                    # mivit_conv_bn
    _fields_ = [(n, c_void_p) for n in ("weight", "gamma", "beta", "running_mean", "running_var")]


class DeepResNetParams(Structure):          # mivit_deepresnet_params
    _fields_ = [("conv", ConvBn * 7), ("fc_weight", c_void_p), ("fc_bias", c_void_p)]


class ConvBnGrad(Structure):                # mivit_conv_bn_grad
    _fields_ = [(n, c_void_p) for n in ("weight", "gamma", "beta")]


class DeepResNetGrads(Structure):           # mivit_deepresnet_grads
    _fields_ = [("conv", ConvBnGrad * 7), ("fc_weight", c_void_p), ("fc_bias", c_void_p)]


# every symbol include/mivit_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "mivit_abi_version": (c_int, []),
    "mivit_last_error": (c_char_p, []),
    "mivit_device_count": (c_int, []),
    "mivit_linear_fwd": (c_int, [c_int, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                 c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "mivit_linear_dgrad": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int64,
                                   c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "mivit_linear_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "mivit_linear_wgrad": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p,
                                   c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "mivit_embed_fwd_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mivit_embed_set_variant": (c_int, [c_int]),
    "mivit_embed_wgrad_bf16_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "mivit_embed_wgrad_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mivit_rowstream_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int64,
                                    c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mivit_rowstream_dgrad": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p,
                                      c_int64, c_void_p, c_int64, c_void_p]),
    "mivit_wavestream_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int64,
                                     c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mivit_wavestream_dgrad": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p,
                                       c_int64, c_void_p, c_int64, c_void_p]),
    "mivit_gemm_dma_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "mivit_gemm_dma_set_variant": (c_int, [c_int]),
    "mivit_gemm_dma_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int64,
                                   c_void_p, c_int64, c_void_p, c_void_p]),
    "mivit_gemm_dma_dgrad": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p,
                                     c_int64, c_void_p, c_int64, c_void_p]),
    "mivit_wgrad_bf16_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "mivit_wgrad_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                 c_size_t, c_void_p]),
    "mivit_deepresnet_eval_supported": (c_int, [c_int, c_int]),
    "mivit_deepresnet_eval_fwd": (c_int, [c_int, c_void_p, c_int, c_int, c_int] + [c_void_p] * 16),
    "mivit_deepresnet_train_supported": (c_int, [c_int, c_int]),
    "mivit_deepresnet_train_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mivit_deepresnet_train_fwd": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p,
                                           c_void_p, c_size_t, c_void_p]),
    "mivit_deepresnet_infer": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_size_t,
                                       c_void_p]),
    "mivit_deepresnet_train_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                           c_void_p, c_size_t, c_void_p]),
    "mivit_deepresnet_train_workspace_layout": (c_int, [c_int, c_int, c_int, c_int, c_void_p]),
    "mivit_deepresnet_train_fwd_stage": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p,
                                                 c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_void_p]),
    "mivit_deepresnet_train_bwd_stage": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                                 c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_void_p]),
    "mivit_graph_stats": (None, [c_void_p, c_void_p, c_void_p]),
    "mivit_wgrad_small_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "mivit_wgrad_small": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                  c_size_t, c_void_p]),
    "mivit_layernorm_fwd": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int64,
                                    c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mivit_layernorm_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mivit_layernorm_bwd": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                    c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int,
                                    c_void_p, c_size_t, c_void_p]),
    "mivit_attention_max_seq": (c_int, [c_int, c_int]),
    "mivit_attention_fwd": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mivit_attention_bwd": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mivit_plan_create": (c_void_p, [POINTER(MivitConfig)]),
    "mivit_plan_destroy": (None, [c_void_p]),
    "mivit_plan_num_params": (c_int, [c_void_p]),
    "mivit_plan_param_name": (c_char_p, [c_void_p, c_int]),
    "mivit_plan_param_offset": (c_int64, [c_void_p, c_int]),
    "mivit_plan_param_numel": (c_int64, [c_void_p, c_int]),
    "mivit_plan_arena_numel": (c_int64, [c_void_p]),
    "mivit_plan_num_stages": (c_int, [c_void_p]),
    "mivit_plan_stage_range": (c_int, [c_void_p, c_int, POINTER(c_int64), POINTER(c_int64)]),
    "mivit_plan_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "mivit_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_int,
                              c_void_p, c_void_p]),
    "mivit_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mivit_fused_layer_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "mivit_attn_block_fwd": (c_int, [c_void_p] * 9 + [c_int, c_int] + [c_void_p] * 8),
    "mivit_mlp_block_fwd": (c_int, [c_void_p] * 9 + [c_int, c_int] + [c_void_p] * 8),
    "mivit_mlp_block_bwd_workspace_bytes": (c_size_t, [c_int]),
    "mivit_mlp_block_bwd_set_waves": (c_int, [c_int]),
    "mivit_mlp_block_bwd": (c_int, [c_void_p] * 10 + [c_int, c_int] + [c_void_p] * 8 + [c_size_t, c_void_p]),
    "mivit_render_frames": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "mivit_attn_out_bwd_workspace_bytes": (c_size_t, [c_int]),
    "mivit_attn_out_bwd": (c_int, [c_void_p] * 6 + [c_int] + [c_void_p] * 7 + [c_size_t, c_void_p]),
    "mivit_embed_small_supported": (c_int, [c_int, c_int, c_int]),
    "mivit_embed_small_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mivit_embed_small_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "mivit_embed_small_wgrad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mivit_qkv_bwd_workspace_bytes": (c_size_t, [c_int]),
    "mivit_qkv_bwd": (c_int, [c_void_p] * 4 + [c_int] + [c_void_p] * 4 + [c_size_t, c_void_p]),
    "mivit_qkv_bwd_workspace_bytes_w64": (c_size_t, [c_int]),
    "mivit_qkv_bwd_w64": (c_int, [c_void_p] * 4 + [c_int] + [c_void_p] * 4 + [c_size_t, c_void_p]),
    "mivit_fused_layer_supported_w64": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "mivit_attn_block_fwd_w64": (c_int, [c_void_p] * 9 + [c_int, c_int] + [c_void_p] * 8),
    "mivit_mlp_block_fwd_w64": (c_int, [c_void_p] * 9 + [c_int, c_int] + [c_void_p] * 8),
    "mivit_mlp_block_bwd_workspace_bytes_w64": (c_size_t, [c_int]),
    "mivit_mlp_block_bwd_w64": (c_int, [c_void_p] * 10 + [c_int, c_int] + [c_void_p] * 8 + [c_size_t, c_void_p]),
    "mivit_attn_out_bwd_workspace_bytes_w64": (c_size_t, [c_int]),
    "mivit_attn_out_bwd_w64": (c_int, [c_void_p] * 6 + [c_int] + [c_void_p] * 7 + [c_size_t, c_void_p]),
    "mivit_profile_enable": (c_int, [ctypes.c_uint64]),
    "mivit_profile_collect": (c_int, [c_int, POINTER(ctypes.c_double), POINTER(c_int)]),
    "mivit_profile_tag_name": (c_char_p, [c_int]),
}
PROF_TAGS = ["embed_fwd", "embed_wgrad", "linear_fwd", "linear_dgrad", "linear_wgrad", "attn_fwd", "attn_bwd",
             "ln_fwd", "ln_bwd", "op", "attn_block_fwd", "mlp_block_fwd", "mlp_block_bwd", "attn_out_bwd", "attn_core_bwd",
             "qkv_wgrad", "qkv_dgrad", "qkv_bwd"]


class MivitError(RuntimeError):
    pass


def _load(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the MiViT HIP extension has not been built. Run "
            "`python -m moleculardiffusion_mivit_amd.csrc.build` (needs hipcc, targets gfx950). "
            "There is no CPU / PyTorch fallback for this path.")
    # PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64.  One process must hold ONE HIP runtime, and device
    # pointers + streams are shared with torch, so torch's copy has to be the one already resident when
    # libmivit_hip.so (NEEDED: libamdhip64.so.7) is resolved: import torch first.
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.mivit_abi_version() != ABI_VERSION:
        raise ImportError(f"libmivit_hip.so ABI {lib.mivit_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
    return lib


lib = _load()

# the same library built with -DMIVIT_STRICT_WAITS (every counted s_waitcnt vmcnt(N) -> vmcnt(0), csrc/build.py).  Test
# infrastructure: tests/test_strict_waits_gpu.py loads it through load_strict() and compares results bitwise; nothing in
# the product does.
STRICT_LIB_PATH = os.path.join(_PKG, "libmivit_hip_strict.so")


def load_strict():
    return _load(STRICT_LIB_PATH)


def check(rc, what=""):
    if rc != 0:
        msg = lib.mivit_last_error()
        raise MivitError(f"{what}: {msg.decode() if msg else 'unknown error'}")


def last_error():
    msg = lib.mivit_last_error()
    return msg.decode() if msg else ""
