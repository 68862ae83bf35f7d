"""CPU: the oracle restatement against the golden vectors the REAL reference produced (tests/golden/*.npz).
This is what keeps the oracle honest on the GPU box, where /root/reference does not exist."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import mivit_oracle as orc
from util import GOLDEN, golden_cases, golden_inputs, load_golden, rel_err, sample_idx

TOL = 2e-5           # oracle vs reference on CPU fp32 (measured <= 1e-5; conv-stack gradients 2e-5)


def _tol(cfg):
    return 1e-4 if cfg.embedding == "deepresnet" else TOL


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_reproduces_reference_outputs_and_grads(name):
    fx, meta, cfg = load_golden(name)
    params, x, labels, feats = golden_inputs(meta, cfg)
    assert torch.equal(labels, torch.from_numpy(fx["labels"]))
    out, loss, grads = orc.loss_and_grads(params, cfg, x, labels, feats, training=meta["training"])
    tol = _tol(cfg)
    assert rel_err(out, fx["out"]) < tol
    assert abs(float(loss) - float(fx["loss"])) / float(fx["loss"]) < tol
    gscale = max(float(np.abs(fx["gsamp/" + k]).max()) for k in grads)
    for k, g in grads.items():
        flat = g.reshape(-1).numpy()
        ref = fx["gsamp/" + k]
        den = float(np.abs(ref).max()) + 1e-3 * gscale
        assert np.abs(flat[sample_idx(flat.size)] - ref).max() / den < tol, k
        assert abs(np.linalg.norm(flat.astype(np.float64)) - float(fx["gnorm/" + k])) / (float(fx["gnorm/" + k]) + 1e-3 * gscale) < tol, k
        if meta["full_grads"]:
            assert np.abs(g.numpy() - fx["grad/" + k]).max() / den < tol, k


@pytest.mark.parametrize("name", ["ref_linear", "ref_cnn", "meanpool_posenc_leaky", "c5_early", "gelu_large_heads"])
def test_oracle_stage_activations(name):
    """Hooked per-stage activations of encoder layer 0 (q/k/v, attention out, norms, FFN) and the final norm."""
    fx, meta, cfg = load_golden(name)
    params, x, labels, feats = golden_inputs(meta, cfg)
    tr = {}
    orc.forward(params, cfg, x, feats, training=meta["training"], trace=tr)
    l0 = tr["layers"][0]
    B, H, S, Dh = l0["q"].shape
    merge = lambda t: t.permute(0, 2, 1, 3).reshape(B, S, H * Dh)   # noqa: E731
    mine = {"embed": tr["embed"], "embed_ln": tr["embed_ln"], "q": merge(l0["q"]), "k": merge(l0["k"]),
            "v": merge(l0["v"]), "attn_out": l0["attn_out"], "x1": l0["x1"], "u": l0["u"], "ffn": l0["ffn"],
            "x2": l0["x2"], "final": tr["final"]}
    for tag, t in mine.items():
        flat = t.reshape(-1).numpy()
        ref = fx["actsamp/" + tag]
        assert np.abs(flat[sample_idx(flat.size)] - ref).max() / (np.abs(ref).max() + 1e-30) < TOL, tag
        if "act/" + tag in fx.files:
            assert rel_err(t, fx["act/" + tag]) < TOL, tag


def test_oracle_batchnorm_running_stats():
    fx, meta, cfg = load_golden("ref_deepresnet_train")
    params, x, labels, _ = golden_inputs(meta, cfg)
    m = orc.OracleModule(cfg, params)
    m.train()
    m(x)
    sd = m.ref_state_dict()
    for k in fx.files:
        if k.startswith("bnstat/"):
            assert rel_err(sd[k[7:]], fx[k]) < 1e-5, k


@pytest.mark.parametrize("name", ["ref_linear", "c1"])
def test_oracle_adamw_trajectory(name):
    """OracleModule + stock AdamW follows the reference's 3-step loss trajectory (pins the CPU-baseline step)."""
    fx, meta, cfg = load_golden(name)
    params, x, labels, _ = golden_inputs(meta, cfg)
    m = orc.OracleModule(cfg, params)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    losses = []
    for _ in range(meta["adamw_steps"]):
        opt.zero_grad()
        loss = F.mse_loss(m(x), labels)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    ref = fx["adamw_losses"]
    assert np.abs(np.array(losses) - ref).max() / np.abs(ref).max() < 1e-5


def test_parameter_count_pins():
    """Known answers printed by the reference's own notebooks / report."""
    with open(os.path.join(GOLDEN, "param_counts.json")) as f:
        pins = json.load(f)
    assert pins["vit_deepresnet_noposenc"]["count"] == 506081
    assert [pins[k]["count"] for k in ("embeddings_small", "embeddings_normal", "embeddings_large")] == [326593, 514273, 1928161]
    for k, v in pins.items():
        cfg = orc.MiViTConfig(**v["config"])
        assert sum(int(np.prod(s)) for s in orc.param_shapes(cfg).values()) == v["count"], k


def test_cnn_embedding_equals_linear_embedding():
    """CNNEmbedding == LinearProjectionEmbedding with the weight viewed (E,1,P,P) (probed 9e-8)."""
    g = torch.Generator().manual_seed(0)
    x = torch.rand(3, 5, 9, 9, generator=g)
    w = torch.randn(16, 81, generator=g)
    b = torch.randn(16, generator=g)
    a = orc.embed_linear(x, w, b)
    c = orc.embed_cnn(x, w.reshape(16, 1, 9, 9), b)
    assert torch.equal(a, c)
    ref = F.conv2d(x.reshape(15, 1, 9, 9), w.reshape(16, 1, 9, 9), b).reshape(3, 5, 16)
    assert rel_err(a, ref) < 1e-6


def test_oracle_edge_cases():
    cfg = orc.MiViTConfig(embedding="linear", patch_size=5, embed_dim=16, num_heads=2, hidden_dim=32, num_layers=1)
    p = orc.closed_form_params(cfg)
    x, y, _ = orc.closed_form_batch(1, 1, 5)             # a single frame, a single sequence
    out = orc.forward(p, cfg, x)
    assert out.shape == (1, 1) and torch.isfinite(out).all()
    assert torch.equal(orc.forward(p, cfg, x[0]), out)   # 3-D input adds the batch dimension
    with pytest.raises(AssertionError, match="Patch size mismatch"):
        orc.forward(p, cfg, torch.zeros(1, 2, 6, 6))
    with pytest.raises(ValueError):
        orc.forward(p, cfg, torch.zeros(1, 2, 3, 5, 5))
    with pytest.raises(AssertionError, match="divisible"):
        orc.attention(torch.zeros(1, 2, 10), {}, "", 3)
