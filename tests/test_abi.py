"""CPU: the C-ABI library loads, exports every symbol include/mivit_hip.h declares, and its host-side logic
(plan / arena layout / workspace sizing / error conventions) is right.  No kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import mivit_oracle as orc
from util import build_product_model, golden_cases, load_golden

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "mivit_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mivit_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from moleculardiffusion_mivit_amd import _native as N
    declared = _declared_symbols()
    assert len(declared) >= 28
    raw = ctypes.CDLL(N.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/mivit_hip.h but not exported"
    assert sorted(N.SYMBOLS) == declared, "ctypes binding table and header disagree"
    assert N.lib.mivit_abi_version() == N.ABI_VERSION == 1
    assert N.lib.mivit_device_count() >= 0


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """No silent fallback: a missing .so makes the binding raise ImportError with build instructions."""
    import importlib.util
    src = os.path.join(ROOT, "moleculardiffusion_mivit_amd", "_native.py")
    dst = tmp_path / "_native_copy.py"
    dst.write_text(open(src).read())
    spec = importlib.util.spec_from_file_location("_native_copy", dst)
    mod = importlib.util.module_from_spec(spec)
    with pytest.raises(ImportError, match="has not been built"):
        spec.loader.exec_module(mod)


@pytest.mark.parametrize("name", golden_cases())
def test_plan_arena_matches_reference_state_dict_schema(name):
    _, meta, cfg = load_golden(name)
    m = build_product_model(cfg, "fp32", None, device="cpu")
    shapes = orc.param_shapes(cfg)
    sd = m.state_dict()
    assert [k for k in sd if not k.endswith("num_batches_tracked") and "running_" not in k] == list(shapes)
    for k, shp in shapes.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    plan = m._plan
    # every transformer-path parameter is in the arena exactly once, 16-byte aligned, no overlaps
    spans = sorted(zip(plan.param_offsets, plan.param_numels, plan.param_names))
    end = 0
    for off, n, nm in spans:
        assert off >= end and (off % 8 == 0 or "k_proj" in nm or "v_proj" in nm), nm
        assert n == int(np.prod(shapes[nm])), nm
        end = off + n
    assert end <= plan.arena_numel
    outside = [k for k in shapes if k not in plan.param_names]
    assert all(k.startswith("embedding.") for k in outside) and (outside == [] or cfg.embedding == "deepresnet")
    # stage ranges tile the arena in backward order: head, layers L-1..0, embedding
    assert plan.num_stages == cfg.num_layers + 2
    prev = 0
    for b, e in plan.stage_ranges:
        assert b == prev and e >= b
        prev = e
    assert prev == plan.arena_numel
    # q/k/v weights are contiguous (one [3E,E] GEMM operand), likewise the biases
    E = cfg.embed_dim
    off = dict(zip(plan.param_names, plan.param_offsets))
    for l in range(cfg.num_layers):
        pre = f"transformer.encoder_layers.{l}.self_attn."
        assert off[pre + "k_proj.weight"] == off[pre + "q_proj.weight"] + E * E
        assert off[pre + "v_proj.weight"] == off[pre + "q_proj.weight"] + 2 * E * E
        assert off[pre + "k_proj.bias"] == off[pre + "q_proj.bias"] + E
    # parameters are views into the arena
    base = m._arena.data_ptr()
    for p, o in zip(m._arena_params, plan.param_offsets):
        assert p.data_ptr() == base + 4 * o


def test_arena_survives_load_state_dict_and_optimizer_step():
    _, meta, cfg = load_golden("ref_linear")
    params = orc.closed_form_params(cfg)
    m = build_product_model(cfg, "fp32", params, device="cpu")
    assert m._arena_ok()
    for k, v in m.state_dict().items():
        assert torch.equal(v, params[k]), k
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    assert m._arena_ok()
    off = dict(zip(m._plan.param_names, m._plan.param_offsets))
    w = m.transformer.encoder_layers[0].self_attn.q_proj.weight
    assert torch.equal(m._arena[off["transformer.encoder_layers.0.self_attn.q_proj.weight"]:][:w.numel()].view_as(w), w)
    import copy
    m2 = copy.deepcopy(m)
    if not m2._arena_ok():
        m2._flatten()
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_workspace_sizing_monotone_and_inference_smaller():
    _, meta, cfg = load_golden("c1")
    m = build_product_model(cfg, "bf16", None, device="cpu")
    p = m._plan
    a = p.workspace_bytes(8, 32, True)
    b = p.workspace_bytes(64, 32, True)
    c = p.workspace_bytes(64, 32, False)
    assert 0 < a < b and c < b
    # once the activations dominate the weight-gradient slabs, fp32 storage needs more than bf16 storage
    m32 = build_product_model(cfg, "fp32", None, device="cpu")
    assert m32._plan.workspace_bytes(2048, 32, True) > p.workspace_bytes(2048, 32, True)


def test_error_conventions_host_side():
    from moleculardiffusion_mivit_amd import _native as N
    from moleculardiffusion_mivit_amd.engine import MivitPlan
    from moleculardiffusion_mivit_amd.helpers import models as M
    kw = dict(precision="fp32", embedding=N.EMBED_LINEAR, patch_size=9, embed_dim=64, num_heads=4, hidden_dim=128,
              num_layers=2, activation=N.ACT_RELU, use_pos_encoding=False, use_regression_token=True,
              fusion=N.FUSION_NONE, global_feature_dim=0, head_hidden=128, output_dim=1)
    MivitPlan(**kw)
    with pytest.raises(N.MivitError, match="divisible by num_heads"):
        MivitPlan(**{**kw, "num_heads": 3})
    with pytest.raises(N.MivitError, match="global_feature_dim"):
        MivitPlan(**{**kw, "fusion": N.FUSION_LATE})
    with pytest.raises(ValueError):
        MivitPlan(**{**kw, "precision": "fp8"})
    # reference-shaped Python errors
    with pytest.raises(AssertionError, match="embed_dim must be divisible by num_heads"):
        M.MultiHeadAttention(10, 3)
    with pytest.raises(ValueError, match="activation_fct must be a callable"):
        M.FeedForward(8, 16, "relu")
    # legal constructor points without fused kernels build, and are routed to the composed path (tests/test_model_gpu.py)
    assert M.FeedForward(8, 16, torch.tanh)._act_code is None
    m = M.GeneralTransformer(M.LinearProjectionEmbedding, dict(patch_size=9, embed_dim=64), 64, 4, 128, 1, M.MLPHead,
                             F.relu, dropout=0.1)
    assert m._composed and m.transformer.encoder_layers[0].dropout.p == 0.1
    with pytest.raises(RuntimeError, match="GPU tensors"):                     # ... which has no CPU fallback either
        m(torch.zeros(1, 3, 9, 9))
    with pytest.raises(AssertionError, match="Must provide global_feature_dim"):
        M.GeneralTransformer(M.LinearProjectionEmbedding, dict(patch_size=9, embed_dim=64), 64, 4, 128, 1, M.MLPHead,
                             F.relu, use_global_features=True)
    # C-ABI argument validation happens before any launch
    rc = N.lib.mivit_linear_fwd(7, None, 0, 0, None, None, 1, 1, 1, 0, None, 0, None, 0, None, None)
    assert rc != 0 and b"dtype" in N.lib.mivit_last_error()
    rc = N.lib.mivit_attention_fwd(N.F32, None, 1, 1, 1, 16, None, None)
    assert rc != 0 and b"null" in N.lib.mivit_last_error()
    assert N.lib.mivit_attention_max_seq(N.BF16, 32) >= 65 and N.lib.mivit_attention_max_seq(N.F32, 64) >= 65


def test_datasets():
    from moleculardiffusion_mivit_amd.helpers.models import ImageDataset, ImageFeatureDataset
    im, ft, lb = torch.arange(24.).reshape(4, 3, 2), torch.arange(8.).reshape(4, 2), torch.arange(4.).reshape(4, 1)
    d = ImageDataset(im, lb)
    assert len(d) == 4 and torch.equal(d[2][0], im[2]) and torch.equal(d[2][1], lb[2])
    d = ImageFeatureDataset(im, ft, lb)
    assert len(d) == 4 and torch.equal(d[1][1], ft[1])
    xb, fb, yb = next(iter(torch.utils.data.DataLoader(d, batch_size=2)))
    assert xb.shape == (2, 3, 2) and fb.shape == (2, 2) and yb.shape == (2, 1)


@pytest.mark.parametrize("P", [0, 5000])
@pytest.mark.parametrize("staged", [False, True])
def test_deepresnet_entries_reject_unsupported_frames_without_crashing(P, staged):
    """A frame side the convolution tiling cannot serve (no frame slot fits a workgroup) must come back as an error
    string -- the launchers divide by the slot count, so every entry checks it first (round-1 record: a host SIGFPE
    from an experimental half-size launcher that skipped the check).  No kernel is launched: validation comes first."""
    from moleculardiffusion_mivit_amd import _native as N
    fake = 0x1000                                   # never dereferenced: the shape check precedes every launch
    prm, gr = N.DeepResNetParams(), N.DeepResNetGrads()
    for i in range(7):
        prm.conv[i] = N.ConvBn(fake, fake, fake, None, None)
        gr.conv[i] = N.ConvBnGrad(fake, fake, fake)
    prm.fc_weight = prm.fc_bias = gr.fc_weight = gr.fc_bias = fake
    vp = ctypes.c_void_p
    assert N.lib.mivit_deepresnet_train_supported(N.BF16, P) == 0
    assert N.lib.mivit_deepresnet_train_workspace_bytes(N.BF16, 4, P, 64) == 0
    if staged:
        rc_f = N.lib.mivit_deepresnet_train_fwd_stage(N.BF16, ctypes.addressof(prm), vp(fake), 4, P, 64, 0.1, 1e-5, vp(fake),
                                                      vp(fake), 1 << 30, 1, vp(fake), vp(fake), None)
        rc_b = N.lib.mivit_deepresnet_train_bwd_stage(N.BF16, ctypes.addressof(prm), vp(fake), vp(fake), 4, P, 64, 1e-5,
                                                      ctypes.addressof(gr), vp(fake), 1 << 30, 1, vp(fake), vp(fake), None)
    else:
        rc_f = N.lib.mivit_deepresnet_train_fwd(N.BF16, ctypes.addressof(prm), vp(fake), 4, P, 64, 0.1, 1e-5, vp(fake),
                                                vp(fake), 1 << 30, None)
        rc_b = N.lib.mivit_deepresnet_train_bwd(N.BF16, ctypes.addressof(prm), vp(fake), vp(fake), 4, P, 64, 1e-5,
                                                ctypes.addressof(gr), vp(fake), 1 << 30, None)
    assert rc_f != 0 and rc_b != 0
    assert "unsupported frame side" in N.last_error()
    with pytest.raises(N.MivitError):
        N.check(rc_b, "mivit_deepresnet_train_bwd")
