"""GPU: the synthetic-data renderer.  The hand-written kernel mivit_render_frames (csrc/render.hip)
against (1) a naive restatement of the reference's loop -- 2-D Gaussians on the up-times finer grid, peak-rescaled, summed,
block-mean pooled (helpers/helpersGeneration.py:283-319) -- and (2) the vectorised CPU path of helpers/generation.py; the noise
terms (clipped-Gaussian background, Poisson gain) through their moments.  The reference generator itself cannot be imported
here or in the build container (helpersGeneration.py:4-6 needs andi_datasets and skimage, both absent and unfetchable), so this
row is pinned by restatement, not by reference outputs."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def naive_frames(traj, npos, sigmas, P, up, amp, center):
    N, T, _ = traj.shape
    F, G = T // npos, P * up
    limit = (G - 1) // 2
    axis = np.linspace(-limit, limit, G)
    out = np.zeros((N, len(sigmas), F, P, P))
    for n in range(N):
        for f in range(F):
            seg = traj[n, f * npos:(f + 1) * npos].astype(np.float64)
            if center:
                seg = seg - seg.mean(axis=0)
            for si, s in enumerate(sigmas):
                hr = np.zeros((G, G))
                for p in range(npos):
                    spot = np.exp(-((axis[None, :] - seg[p, 0] * up) ** 2 + (axis[:, None] - seg[p, 1] * up) ** 2) / (2 * s * s))
                    hr += amp[n, f, p] / spot.max() * spot          # float64, as the reference: no floor on spot_max (:305-308)
                out[n, si, f] = hr.reshape(P, up, P, up).mean(axis=(1, 3))
    return out


@pytest.mark.parametrize("N,T,npos,P,up,center", [(3, 20, 5, 9, 5, True), (2, 12, 3, 8, 4, False), (2, 60, 10, 13, 5, True),
                                                  (1, 8, 2, 64, 5, True)])
def test_render_kernel_matches_naive_loop_and_cpu_path(N, T, npos, P, up, center):
    from moleculardiffusion_mivit_amd.helpers import generation as gen
    g = torch.Generator().manual_seed(N * 100 + T)
    traj = torch.randn(N, T, 2, generator=g) * 1.2
    traj[0, :npos] += P                       # a particle that left the frame: the reference still rescales its peak to the
                                              # full intensity, which then sits on the border ("Particle Left the image")
    amp = 500 + 50 * torch.randn(N, T // npos, npos, generator=g)
    sigmas = [2.3, 1.1] if P <= 13 else [4.0]
    cpu = gen.render_frames(traj, npos, sigmas, P, up, amp, center)
    hip = gen.render_frames(traj.cuda(), npos, sigmas, P, up, amp.cuda(), center)
    torch.cuda.synchronize()
    assert hip.is_cuda and hip.shape == cpu.shape
    scale = float(cpu.abs().max())
    assert float((hip.cpu() - cpu).abs().max()) < 2e-5 * scale
    if P <= 13:
        ref = naive_frames(traj.numpy(), npos, sigmas, P, up, amp.numpy(), center)
        assert float(np.abs(hip.cpu().numpy() - ref).max()) < 2e-5 * scale


def test_noise_terms_have_the_reference_moments_on_gpu():
    """Background np.clip(N(bm, bs), 0, bm + 3 bs) and the Poisson gain poisson(pn) / pn (helpersGeneration.py:312-317)."""
    from moleculardiffusion_mivit_amd.helpers import generation as gen
    g = torch.Generator(device="cuda").manual_seed(0)
    bm, bs = 1420.0, 290.0
    bg = gen.clipped_background((400, 64, 64), bm, bs, g, "cuda")
    assert float(bg.min()) >= 0 and float(bg.max()) <= bm + 3 * bs + 1e-3
    # clipping at +3 sigma only: mean shifts by -sigma * (phi(3) - 3 (1 - Phi(3))) ~ -1.1e-4 sigma... well inside the noise
    assert abs(float(bg.mean()) - bm) < 0.01 * bs and abs(float(bg.std()) - bs) < 0.01 * bs
    props = dict(gen.DEFAULT_IMAGE_PROPS)
    props.update({"particle_intensity": [0.0, 0.0], "background_intensity": [bm, 0.0], "poisson_noise": 100, "output_size": 16})
    traj = torch.zeros(200, 30, 2, device="cuda")
    vid = gen.trajectories_to_video(traj, 10, image_props=props, generator=g, device="cuda")
    # no particle, constant background: pixel = bm * Poisson(100) / 100 -> mean bm, std bm / sqrt(100)
    assert vid.shape == (200, 3, 16, 16)
    assert abs(float(vid.mean()) - bm) < 0.002 * bm and abs(float(vid.std()) - bm / 10) < 0.02 * bm / 10


# ---- SURVEY section 8 row f4 on the GPU: real-data patch extraction and the ResNet comparison baselines -------------------
def test_patch_extraction_on_gpu_tensors_matches_cpu():
    """helpers/tracking.extract_particle_patches (reference helpersTracking.py:513-550) on a GPU movie: the gather runs on the
    device and returns device tensors equal to the CPU (numpy) result, border positions zero-padded."""
    from moleculardiffusion_mivit_amd.helpers.tracking import extract_particle_patches
    g = torch.Generator().manual_seed(5)
    movie = torch.rand(12, 40, 52, generator=g)
    tracks = {0: [(0, 3.2, 4.7), (1, 0.4, 51.6), (2, 39.5, 0.2)], "b": [(t, 20.0 + 1.3 * t, 25.0 - 0.9 * t) for t in range(12)], 7: []}
    want = extract_particle_patches(movie.numpy(), tracks, patch_size=9)
    got = extract_particle_patches(movie.cuda(), tracks, patch_size=9)
    for k in tracks:
        assert got[k].is_cuda
        assert got[k].shape[0] == len(tracks[k])
        if len(tracks[k]):
            assert torch.equal(got[k].cpu(), torch.as_tensor(want[k]))


@pytest.mark.parametrize("kind", ["images", "images_features"])
def test_resnet_baselines_train_on_gpu_like_on_cpu(kind):
    """MultiImageResNet / MultiImageFeatureResNet (reference helpers/models.py:600-772; NOT on the MiViT path: stock PyTorch-ROCm
    modules kept so that every factory of getTrainingModels builds): forward, loss and one AdamW step on the GPU track the CPU run
    of the same module (MIOpen convolutions vs CPU: 1e-3)."""
    import torch.nn.functional as F
    from moleculardiffusion_mivit_amd.helpers.models import MultiImageFeatureResNet, MultiImageResNet
    torch.manual_seed(3)
    m = MultiImageResNet(9) if kind == "images" else MultiImageFeatureResNet(9, 25)
    m.train()
    x, f, y = torch.rand(6, 30, 9, 9), torch.randn(6, 25), torch.rand(6, 1)
    import copy
    mg = copy.deepcopy(m).cuda()
    outs = []
    for mod, dev in ((m, "cpu"), (mg, "cuda")):
        opt = torch.optim.AdamW(mod.parameters(), lr=1e-3)
        args = (x.to(dev),) if kind == "images" else (x.to(dev), f.to(dev))
        out = mod(*args)
        loss = F.mse_loss(out, y.to(dev))
        loss.backward()
        opt.step()
        outs.append((out.detach().cpu(), float(loss.detach()), mod(*args).detach().cpu()))
    (o0, l0, s0), (o1, l1, s1) = outs
    assert o1.shape == (6, 1)
    assert float((o0 - o1).abs().max()) < 1e-3 * max(1.0, float(o0.abs().max())) and abs(l0 - l1) < 1e-3 * max(1.0, abs(l0))
    assert float((s0 - s1).abs().max()) < 5e-2 * max(1.0, float(s0.abs().max()))       # after one optimizer step (BatchNorm batch statistics of 6 sequences)
