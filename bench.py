#!/usr/bin/env python3
"""MiViT training-throughput benchmark (BASELINE.json metric: training image-sequences/sec).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = zero_grad + forward + MSELoss + backward (+ per-stage gradient all-reduce over RCCL when N > 1) +
AdamW.step, on the PSFNoise 32-frame 64x64 configuration (ViT depth 4, dim 128, 4 heads, hidden 256, linear frame
embedding, regression token) -- BASELINE.json configs[1] -- with synthetic sequences already resident in HBM.
Weak scaling: the per-GPU batch is fixed, each rank draws its own shard (seed + rank).
Prints ONE JSON line on rank 0 (see the contract in the task prompt), including `roofline` for the dominant
kernel (timed with hipEvents inside the library over the timed region) and `cpu_baseline` (the oracle's plain
PyTorch CPU restatement timed on this host, rank 0, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK = {"bf16": 2500.0, "fp32": 157.3}   # dense TFLOP/s

CFG = dict(patch_size=64, embed_dim=128, num_heads=4, hidden_dim=256, num_layers=4, frames=32)


def synth(B, T, P, seed, device):
    """Synthetic PSF-blob sequences generated ON the GPU: background N(0.2, 0.06^2) + one Gaussian spot
    doing Brownian motion with D ~ U(0.1, 10); label D / 10."""
    g = torch.Generator(device=device).manual_seed(seed)
    D = torch.rand(B, generator=g, device=device) * 9.9 + 0.1
    steps = torch.randn(B, T, 2, generator=g, device=device) * torch.sqrt(2 * D * 0.01).view(B, 1, 1) * (P / 9.0)
    pos = torch.cumsum(steps, dim=1)
    pos = pos - pos.mean(dim=1, keepdim=True) + (P - 1) / 2.0
    yy = torch.arange(P, dtype=torch.float32, device=device).view(1, 1, P, 1)
    xx = torch.arange(P, dtype=torch.float32, device=device).view(1, 1, 1, P)
    sig = 1.1 * P / 9.0
    x = torch.randn(B, T, P, P, generator=g, device=device) * 0.06 + 0.2
    x += 0.6 * torch.exp(-((yy - pos[..., 1].view(B, T, 1, 1)) ** 2 + (xx - pos[..., 0].view(B, T, 1, 1)) ** 2)
                         / (2 * sig * sig))
    return x.contiguous(), (D / 10.0).view(B, 1).contiguous()


def flops_per_seq(T, P, E, H, Fh, L, head_hidden=128):
    S = T + 1
    embed = 2 * T * P * P * E
    layer = 6 * S * E * E + 4 * S * S * E + 2 * S * E * E + 4 * S * E * Fh
    head = 2 * E * head_hidden + 2 * head_hidden
    return {"embed_fwd": embed, "attn_mlp_fwd": L * layer, "total_fwd": embed + L * layer + head}


def cpu_baseline(seconds=12.0):
    """The oracle's plain-PyTorch fp32 restatement of the same step on the host cores (kind = 'port')."""
    from oracle import mivit_oracle as orc
    cfg = orc.MiViTConfig(embedding="linear", patch_size=CFG["patch_size"], embed_dim=CFG["embed_dim"],
                          num_heads=CFG["num_heads"], hidden_dim=CFG["hidden_dim"], num_layers=CFG["num_layers"])
    torch.manual_seed(0)
    m = orc.OracleModule(cfg, seed=0)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    B = 32
    x, y, _ = orc.synthetic_batch(B, CFG["frames"], CFG["patch_size"], seed=1234)

    def step():
        opt.zero_grad()
        loss = F.mse_loss(m(x), y)
        loss.backward()
        opt.step()
    for _ in range(2):
        step()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        step()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(n * B / dt, 2), "unit": "sequences/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} steps of batch {B} ({n * B} sequences, {dt:.1f} s) of the same 32x64x64 depth-4 dim-128 "
                      f"train step, oracle/mivit_oracle.py on torch CPU fp32"}


def launch_ranks(n):
    """Start `n` ranks of this script (one per GPU) under torch.distributed.run and wait for them."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL needs it on this pool)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=int(os.environ.get("MIVIT_BENCH_BATCH", 16384)))
    ap.add_argument("--precision", default=os.environ.get("MIVIT_BENCH_PRECISION", "bf16"), choices=["bf16", "fp32"])
    ap.add_argument("--resident-batches", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the B=4096 / fp32-parity-mode / trained-val side runs")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel-category time table to stderr")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process has not touched the GPU (importing torch does not) and never
        # will -- it starts the N ranks through torch.distributed.run as a CHILD process, relays their output (rank 0
        # prints the JSON line) and exits with the children's return code.
        sys.exit(launch_ranks(args.gpus))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    import torch.distributed as dist
    # rehearsal knobs (not used by the driver): several ranks on ONE GPU over gloo, to exercise the launcher path
    if os.environ.get("MIVIT_BENCH_SHARE_GPU") == "1":
        local = 0
    backend = os.environ.get("MIVIT_DIST_BACKEND", "nccl")       # "nccl" IS RCCL on ROCm
    if os.environ.get("MIVIT_BENCH_DRY") == "1":
        # launcher rehearsal without a GPU (tests/test_bench_launcher.py): rendezvous over gloo, one collective, one line
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(tt)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": dist.get_world_size(), "rank_sum": float(tt.item()),
                              "steps": args.steps, "warmup": args.warmup}), flush=True)
        dist.destroy_process_group()
        return
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from moleculardiffusion_mivit_amd import _native as N
    from moleculardiffusion_mivit_amd import dp
    from moleculardiffusion_mivit_amd.helpers.models import GeneralTransformer, LinearProjectionEmbedding, MLPHead

    T, P, E, H, Fh, L = (CFG["frames"], CFG["patch_size"], CFG["embed_dim"], CFG["num_heads"], CFG["hidden_dim"],
                         CFG["num_layers"])
    loss_fn = torch.nn.MSELoss()

    def build(precision, lr=1e-4, seed=0):
        torch.manual_seed(seed)
        m = GeneralTransformer(LinearProjectionEmbedding, {"patch_size": P, "embed_dim": E}, E, H, Fh, L, MLPHead,
                               F.relu, dropout=0.0, use_pos_encoding=False, use_regression_token=True,
                               precision=precision).to(dev)
        m.train()
        if world > 1:
            dp.attach(m)
        return m, torch.optim.AdamW(m.parameters(), lr=lr, fused=True)

    def make_step(m, opt, data):
        def step(i):
            x, y = data[i % len(data)]
            opt.zero_grad(set_to_none=True)
            loss = loss_fn(m(x), y)
            loss.backward()
            opt.step()
            return loss
        return step

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps, warmup):
        for i in range(warmup):
            step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        barrier()
        return time.perf_counter() - t0

    model, opt = build(args.precision)
    Bg = args.batch_per_gpu
    data = [synth(Bg, T, P, 1234 + rank + 1000 * i, dev) for i in range(args.resident_batches)]
    step = make_step(model, opt, data)

    # ---- warmup; the last warmup steps time every kernel category ----
    nprof = min(2, args.warmup)
    for i in range(args.warmup - nprof):
        step(i)
    torch.cuda.synchronize()
    N.lib.mivit_profile_enable(ctypes.c_uint64((1 << len(N.PROF_TAGS)) - 1))
    for i in range(nprof):
        step(i)
    torch.cuda.synchronize()
    cat = {}
    for t, name in enumerate(N.PROF_TAGS):
        ms, cnt = ctypes.c_double(), ctypes.c_int()
        N.lib.mivit_profile_collect(t, ctypes.byref(ms), ctypes.byref(cnt))
        cat[name] = (ms.value / max(nprof, 1), cnt.value // max(nprof, 1))
    # the kernel that owns the roofline entry: the single launch with the most time per step among the HBM-streaming ones
    # (the frame-embedding forward); the encoder-layer family, which owns the largest SHARE of the step, gets `roofline_layers`
    dominant = "embed_fwd"
    if args.breakdown and rank == 0:
        tot = sum(v[0] for v in cat.values())
        for k, (ms, c) in sorted(cat.items(), key=lambda kv: -kv[1][0]):
            print(f"  {k:14s} {ms:9.3f} ms/step  {c:4d} launches/step  {100 * ms / max(tot, 1e-9):5.1f}%", file=sys.stderr)
    dom_tag = N.PROF_TAGS.index(dominant)
    layer_tags = ("linear_fwd", "linear_dgrad", "linear_wgrad", "attn_fwd", "attn_bwd", "ln_fwd", "ln_bwd", "attn_block_fwd",
                  "mlp_block_fwd", "mlp_block_bwd", "attn_out_bwd", "attn_core_bwd", "qkv_wgrad", "qkv_dgrad", "qkv_bwd")
    N.lib.mivit_profile_enable(ctypes.c_uint64(1 << dom_tag))

    # ---- timed region: EXACTLY --steps steps ----
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms, cnt = ctypes.c_double(), ctypes.c_int()
    N.lib.mivit_profile_collect(dom_tag, ctypes.byref(ms), ctypes.byref(cnt))
    N.lib.mivit_profile_enable(ctypes.c_uint64(0))

    # ---- val MSE(D) of the (untrained, K-step) benchmark model on a held-out synthetic shard ----
    model.eval()
    with torch.no_grad():
        xv, yv = synth(min(Bg, 1024), T, P, 99991 + rank, dev)
        val_mse = float(F.mse_loss(model(xv) * 10.0, yv * 10.0))
    del model, opt, data, step
    torch.cuda.empty_cache()

    extras = {}
    if world == 1 and not args.no_extras:
        # (1) the SURVEY 8d batch point 4096 and the fp32 parity mode (v_mfma_f32_16x16x4_f32 kernels), short runs
        for tag, prec, bsz in (("batch_4096", args.precision, 4096), ("fp32_parity_mode", "fp32", 2048)):
            m2, o2 = build(prec)
            d2 = [synth(bsz, T, P, 777 + i, dev) for i in range(2)]
            t2 = timed(make_step(m2, o2, d2), 8, 3)
            extras[tag] = {"sequences_per_s": round(8 * bsz / t2, 1), "ms_per_step": round(1e3 * t2 / 8, 4), "per_gpu_batch": bsz,
                           "dtype": prec}
            del m2, o2, d2
            torch.cuda.empty_cache()
        # (1b) inference (model.eval() + no_grad: the make_prediction path of the reference's evaluation loops): forward only,
        #      the layers hand each other normalised tokens and nothing is kept for a backward
        m2, _ = build(args.precision)
        m2.eval()
        d2 = [synth(Bg, T, P, 888 + i, dev) for i in range(2)]
        with torch.no_grad():
            for i in range(3):
                m2(d2[i % 2][0])
            barrier()
            t0i = time.perf_counter()
            for i in range(10):
                m2(d2[i % 2][0])
            barrier()
            t2 = time.perf_counter() - t0i
        extras["inference"] = {"sequences_per_s": round(10 * Bg / t2, 1), "ms_per_batch": round(1e3 * t2 / 10, 4), "per_gpu_batch": Bg,
                               "dtype": args.precision}
        del m2, d2
        torch.cuda.empty_cache()
        # (2) a TRAINED validation MSE(D): fixed short schedule (AdamW 1e-3, 300 steps of 1024 fresh synthetic sequences,
        #     StepLR(5, 0.9) every 30 steps), same seed in the benchmark precision and in the fp32 parity mode
        for tag, prec in (("val_mse_D_trained", args.precision), ("val_mse_D_trained_fp32", "fp32")):
            m3, o3 = build(prec, lr=1e-3, seed=1)
            sch = torch.optim.lr_scheduler.StepLR(o3, step_size=5, gamma=0.9)
            for s_ in range(300):
                x3, y3 = synth(1024, T, P, 50000 + s_, dev)
                o3.zero_grad(set_to_none=True)
                loss_fn(m3(x3), y3).backward()
                o3.step()
                if (s_ + 1) % 30 == 0:
                    sch.step()
            m3.eval()
            with torch.no_grad():
                xv3, yv3 = synth(4096, T, P, 99991, dev)
                extras[tag] = round(float(F.mse_loss(m3(xv3) * 10.0, yv3 * 10.0)), 4)
            del m3, o3
            torch.cuda.empty_cache()

    if rank == 0:
        seqs = args.steps * Bg * world
        fl = flops_per_seq(T, P, E, H, Fh, L)
        k_ms = ms.value / max(cnt.value, 1)
        ts = 2 if args.precision == "bf16" else 4
        roof = None
        if cnt.value:
            # algorithmic bytes per launch: the fp32 frames of the per-GPU batch, read once, + weight + output
            byt = Bg * T * P * P * 4 + E * P * P * 4 + Bg * T * E * ts
            ach = byt / (k_ms * 1e-3) / 1e9
            roof = {"kernel": dominant, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "launch_ms": round(k_ms, 4), "launches_timed": cnt.value,
                    "algorithmic_bytes_per_launch": byt}
        # encoder-layer family (everything between the embedding LayerNorm and the final norm, forward + backward): HBM-bound
        # at these widths.  Two byte counts, in units of U = one [tokens, E] activation:
        #  * design bytes -- what THIS design's launches must move given what it keeps for the backward: per layer forward 8 U
        #    (attention block: x in, n1 + ctx + q|k|v out; feed-forward block: n1 in, n2 out), backward 25 U (feed-forward
        #    block 4; LayerNorm-1 + out-projection 5; attention core 7; q|k|v wgrad 4 + dgrad 5) = 33 U;
        #  * algorithmic floor -- inputs, outputs and the minimum saved set under full recompute: a layer reads x and writes
        #    its output in the forward (2 U, the output being the next layer's saved input), and reads x and dy and writes dx
        #    in the backward (3 U): 5 U per layer.  The gap between the two is activations this design saves instead of
        #    recomputing; `frac_of_floor` is the honest distance from the roofline.
        # The byte model belongs to the fused path (bf16, E 128 / F 256 / 4 heads, S <= 64): other paths report times only.
        S = T + 1
        U = Bg * S * E * ts
        Mtok = Bg * S
        fused_path = bool(args.precision == "bf16" and N.lib.mivit_fused_layer_supported(N.BF16, E, Fh, H, S)
                          and not os.environ.get("MIVIT_NO_FUSED_LAYER"))
        layer_ms = sum(cat[k][0] for k in layer_tags if k in cat)
        roof_layers = None
        if layer_ms > 0:
            roof_layers = {"kernels": [k for k in layer_tags if cat.get(k, (0, 0))[0] > 0], "path": "fused" if fused_path else "general",
                           "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "ms_per_step": round(layer_ms, 4),
                           "share_of_step": round(layer_ms / (1e3 * dt / args.steps), 3), "traffic": None}
            if fused_path:
                # 33 U per layer as two q|k|v backward launches (9 U), 30 U with the one-pass kernel (6 U)
                nu_layer = 30 if cat.get("qkv_bwd", (0.0, 0))[0] > 0 else 33
                design, floor = L * nu_layer * U, L * 5 * U
                roof_layers.update({
                    "achieved": round(design / (layer_ms * 1e-3) / 1e9, 1), "frac": round(design / (layer_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "algorithmic_bytes_per_step": design, "bytes_model": f"design: {nu_layer} U per layer (what this design saves and re-reads)",
                    "algorithmic_floor_bytes_per_step": floor, "floor_model": "5 U per layer (x, out | x, dy, dx; everything else recomputed)",
                    "frac_of_floor": round(floor / (layer_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        # per-kernel table of the fused path: ms per step, design bytes, algorithmic FLOPs, fraction of both roofs
        fused_kernels = None
        if fused_path:
            spec = {   # name: (U per layer, FLOPs per token and layer)
                "attn_block_fwd": (6, 8 * E * E + 4 * S * E), "mlp_block_fwd": (2, 4 * E * Fh), "mlp_block_bwd": (4, 10 * E * Fh),
                "attn_out_bwd": (5, 4 * E * E), "attn_core_bwd": (7, 10 * S * E), "qkv_wgrad": (4, 6 * E * E), "qkv_dgrad": (5, 6 * E * E),
                "qkv_bwd": (6, 12 * E * E)}
            fused_kernels = {}
            for k, (nu, fl_tok) in spec.items():
                ms_k = cat.get(k, (0.0, 0))[0]
                if ms_k <= 0:
                    continue
                byt_k, fl_k = L * nu * U, L * fl_tok * Mtok
                fused_kernels[k] = {"ms": round(ms_k, 4), "bytes": byt_k, "flops": fl_k,
                                    "frac_hbm": round(byt_k / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "frac_mfma": round(fl_k / (ms_k * 1e-3) / 1e12 / MFMA_PEAK["bf16"], 4)}
            if fused_kernels:
                worst = min(fused_kernels, key=lambda k: max(fused_kernels[k]["frac_hbm"], fused_kernels[k]["frac_mfma"]))
                fused_kernels["furthest_below_its_roof"] = worst
        # HBM traffic from the PMC passes committed under profiles/ (rocprofv3 cannot run inside this process): used only
        # when it was collected at this per-GPU batch
        for fname in ("r03_pmc.json", "r02_pmc.json", "r01_pmc_embed.json"):
            try:
                with open(os.path.join(ROOT, "profiles", fname)) as fh:
                    pmc = json.load(fh)
                if pmc.get("per_gpu_batch") == Bg and args.precision == "bf16":
                    if roof is not None and roof["traffic"] is None and dominant in pmc.get("kernels", {}):
                        roof["traffic"] = pmc["kernels"][dominant]["hbm_bytes_per_launch"]
                        roof["traffic_source"] = "profiles/" + fname
                    if roof_layers is not None and fused_path and roof_layers["traffic"] is None and "layers_hbm_bytes_per_step" in pmc:
                        roof_layers["traffic"] = pmc["layers_hbm_bytes_per_step"]
                        roof_layers["traffic_source"] = "profiles/" + fname
            except (OSError, ValueError, KeyError):
                pass
        line = {
            "metric": "training image-sequences/sec", "value": round(seqs / dt, 1), "unit": "sequences/s",
            "n_gpus": world, "rccl_ranks": (dist.get_world_size() if world > 1 and backend == "nccl" else 0),
            "dist_backend": (backend if world > 1 else None), "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": "PSFNoise 32-frame 64x64 sequences, MiViT depth=4 dim=128 heads=4 hidden=256, linear "
                                   "frame embedding + regression token; train step = fwd + MSE + bwd + AdamW",
                       "per_gpu_batch": Bg, "global_batch": Bg * world, "frames": T, "frame": f"{P}x{P}",
                       "parallelism": f"dp{world}", "input_dtype": "f32 frames resident in HBM"},
            "val_mse_D": round(val_mse, 4),
            "model_tflops": round(3 * fl["total_fwd"] * seqs / dt / 1e12, 2),
            "attn_mlp_mfma_frac": round(3 * fl["attn_mlp_fwd"] * seqs / dt / 1e12 / world / MFMA_PEAK[args.precision], 5),
            "roofline": roof,
            "roofline_layers": roof_layers,
            "fused_layer_kernels": fused_kernels,
            "kernel_ms_per_step": {k: round(v[0], 4) for k, v in cat.items()},
        }
        line.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
