"""Synthetic-data producer: the vectorised renderer against a naive per-pixel restatement of the
reference's loops (helpersGeneration.py:77-97 gaussian_2d, :283-308 frame synthesis incl. peak re-normalisation and
5x5 block mean), Brownian statistics, noise moments, normalisation formula."""
import math

import numpy as np
import pytest
import torch

from moleculardiffusion_mivit_amd.helpers import generation as gen


def naive_frames(traj_px, npos, sigma, P, up, amps, center):
    """Checker: literal loops (grid, 2-D Gaussian, divide by its max, accumulate, block-mean)."""
    N, T, _ = traj_px.shape
    F_ = T // npos
    G = P * up
    limit = (G - 1) // 2
    ax = np.linspace(-limit, limit, G)
    X, Y = np.meshgrid(ax, ax)
    out = np.zeros((N, F_, P, P))
    for n in range(N):
        for f in range(F_):
            seg = traj_px[n, f * npos:(f + 1) * npos].copy()
            if center:
                seg = seg - seg.mean(axis=0)
            hr = np.zeros((G, G))
            for p in range(npos):
                xc, yc = seg[p, 0] * up, seg[p, 1] * up
                spot = amps[n, f, p] * np.exp(-(((X - xc) ** 2) / (2 * sigma ** 2) + ((Y - yc) ** 2) / (2 * sigma ** 2)))
                hr += amps[n, f, p] / spot.max() * spot
            out[n, f] = hr.reshape(P, up, P, up).mean(axis=(1, 3))
    return out


@pytest.mark.parametrize("P,up,center", [(9, 5, True), (13, 5, False), (8, 4, True)])
def test_render_matches_naive_loops(P, up, center):
    g = torch.Generator().manual_seed(1)
    N, npos, F_ = 2, 4, 3
    traj = torch.randn(N, npos * F_, 2, generator=g).double() * 0.7
    amps = (torch.rand(N, F_, npos, generator=g).double() + 0.5) * 100
    sig = 3.3
    got = gen.render_frames(traj, npos, [sig, sig / 2], P, up, amps, center)
    for i, s in enumerate((sig, sig / 2)):
        ref = naive_frames(traj.numpy(), npos, s, P, up, amps.numpy(), center)
        assert np.abs(got[:, i].numpy() - ref).max() < 1e-9 * ref.max()


def test_brownian_statistics():
    g = torch.Generator().manual_seed(2)
    tr, lab = gen.brownian_single_state(4000, 50, Ds=[2.0, 0.0], generator=g)
    assert tr.shape == (50, 4000, 2) and lab.shape == (50, 4000, 3)
    assert torch.all(lab[..., 0] == 1) and torch.all(lab[..., 1] == 2.0) and torch.all(lab[..., 2] == 0)
    msd = (tr[-1] - tr[0]).pow(2).sum(-1).mean().item()          # 2-D free diffusion: MSD = 4 D t
    assert abs(msd - 4 * 2.0 * 49) / (4 * 2.0 * 49) < 0.06
    tr, lab = gen.brownian_single_state(5000, 3, Ds=[3.0, 1.0], generator=g)
    D = lab[0, :, 1]
    assert (D > 0).all() and abs(D.mean().item() - 3.0) < 0.06 and abs(D.std().item() - 1.0) < 0.06


def test_video_noise_model_moments_and_shapes():
    g = torch.Generator().manual_seed(3)
    tr, _ = gen.brownian_single_state(16, 40, Ds=[1.0, 0.0], generator=g)
    props = {"output_size": 9, "particle_intensity": [0, 0], "background_intensity": [100, 10], "poisson_noise": -1}
    v = gen.trajectories_to_video(tr.permute(1, 0, 2), 10, True, props, generator=g)
    assert v.shape == (16, 4, 9, 9) and v.dtype == torch.float32
    assert abs(v.mean().item() - 100) < 1.0 and abs(v.std().item() - 10) < 1.0     # no particle: pure background
    assert v.max() <= 130.0 + 1e-3 and v.min() >= 0
    props["poisson_noise"] = 100
    v2 = gen.trajectories_to_video(tr.permute(1, 0, 2), 10, True, props, generator=g)
    assert abs(v2.mean().item() - 100) < 1.5 and v2.std().item() > v.std().item()  # multiplicative Poisson(100)/100
    with pytest.raises(Exception, match="divisble"):
        gen.trajectories_to_video(tr.permute(1, 0, 2), 7, True, props)


def test_normalize_images_formula():
    x = torch.arange(12.).reshape(3, 4)
    out, (m, s, mx) = gen.normalize_images(x, 2.0, 1.0, 9.0)
    assert torch.allclose(out, (x - 1.0) / 8.0)
    out, stats = gen.normalize_images(x)
    assert math.isclose(stats[0], 5.5) and math.isclose(stats[2], 11.0)
    assert float(gen.normalize_images(x * 10, 2.0, 1.0, 9.0, clip_image=True)[0].max()) == 1.5
    with pytest.raises(ValueError):
        gen.normalize_images(x, 5.0, 1.0, 4.0)


def test_psfnoise_settings_surface():
    """The reference's settings module surface (trainSettingsPSFNoise.py:9-193): names, model zoo keys, shapes."""
    from moleculardiffusion_mivit_amd.experiments.PSFNoise import trainSettingsPSFNoise as S
    for nm in ("device sequences center adaptive_batch_size lr D_max_normalization loss_function val_loss_function "
               "single_prediction use_regression_token use_pos_encoding tr_activation_fct patch_size embed_dim num_heads "
               "hidden_dim num_layers dropout traj_div_factor nPosPerFrame nFrames T image_props PSF_Settings "
               "Noise_Settings N_PSF N_Noise").split():
        assert hasattr(S, nm), nm
    assert (S.patch_size, S.embed_dim, S.num_heads, S.hidden_dim, S.num_layers, S.nFrames) == (9, 64, 4, 128, 6, 30)
    models, opts, scheds = S.getTrainingModels(psf_indices=[0, 4], noise_indices=[0])
    assert list(models) == ["tr_0_0", "res_0_0", "tr_4_0", "res_4_0"]
    assert sum(p.numel() for p in models["tr_0_0"].parameters() if p.requires_grad) == 506081   # reference notebook pin
    assert sum(p.numel() for p in models["res_0_0"].parameters() if p.requires_grad) == 315617
    assert isinstance(opts["tr_0_0"], torch.optim.AdamW) and scheds["tr_0_0"].step_size == 5
    assert S.select_models_from_psf(models, 4, "tr") == ["tr_4_0"]
    assert S.select_models_from_noise(models, 0) == list(models)
    g = torch.Generator().manual_seed(0)
    tr, _ = gen.brownian_single_state(3, S.T, Ds=[3, 1], generator=g)
    v = S.trajs_to_vid_psf_noise(tr.permute(1, 0, 2).numpy() / S.traj_div_factor, S.nPosPerFrame, center=S.center,
                                 image_props=S.image_props, PSF_Settings=S.PSF_Settings, Noise_Settings=S.Noise_Settings,
                                 generator=g)
    assert v.shape == (3, S.N_PSF, S.N_Noise, S.nFrames, 9, 9) and v.dtype == np.float32
    # PSF_Settings divide sigma; spots are peak-normalised on the fine grid, so the narrowest PSF (index 0,
    # sigma / 2) carries the least integrated flux after pooling
    assert v[:, 0, 0].sum() < v[:, 4, 0].sum()
    # noise grows with the noise index
    assert v[:, 2, 5].std() > v[:, 2, 0].std()
    # the reference's chain (trainSettingsPSFNoise.py:296-306): level 0 = Poisson(clean + bm) overwrites out[psf, 0, f] and the
    # later levels are built ON it, so their background is ~2 * bm; the clean-frame variant has ~bm at every level
    bm = S.image_props["background_intensity"][0]
    corner = lambda a, j: float(a[:, :, j, :, 0, 0].mean())           # noqa: E731  (a pixel far from the spot)
    assert abs(corner(v, 0) - bm) < 0.1 * bm and abs(corner(v, 2) - 2 * bm) < 0.15 * bm
    v1 = S.trajs_to_vid_psf_noise(tr.permute(1, 0, 2).numpy() / S.traj_div_factor, S.nPosPerFrame, center=S.center,
                                  image_props=S.image_props, PSF_Settings=S.PSF_Settings, Noise_Settings=S.Noise_Settings,
                                  generator=torch.Generator().manual_seed(1), reference_chain=False)
    assert abs(corner(v1, 0) - bm) < 0.1 * bm and abs(corner(v1, 2) - bm) < 0.15 * bm


def test_other_experiment_settings_surfaces():
    """Model zoos of the Embeddings / Framerate / ImagesFeatures experiments: key names and the parameter counts the
    reference's report prints (ProjectReport Table 1: 326k / 514k / 1.93M)."""
    from moleculardiffusion_mivit_amd.experiments.Embeddings import trainSettingsEmbeddings as E
    from moleculardiffusion_mivit_amd.experiments.Framerate import trainSettingsFramerate as Fr
    from moleculardiffusion_mivit_amd.experiments.ImagesFeatures import trainSettingsImagesFeatures as IF
    m, o, s = E.getTrainingModels()
    assert list(m) == ["linear_n", "cnn_n", "deepcnn_n", "linear_s", "cnn_s", "deepcnn_s", "linear_b", "cnn_b", "deepcnn_b", "resnet"]
    count = lambda k: sum(p.numel() for p in m[k].parameters() if p.requires_grad)   # noqa: E731
    assert (count("deepcnn_s"), count("deepcnn_n"), count("deepcnn_b")) == (326593, 514273, 1928161)
    assert E.use_pos_encoding and m["linear_n"].transformer.use_pos_encoding
    m, o, s = Fr.getTrainingModels(indices=[0, 5])
    assert list(m) == ["tr_0", "res_0", "tr_5", "res_5"] and Fr.nPosPerFrame_FramesNumber == [60, 30, 20, 15, 10, 6]
    g = torch.Generator().manual_seed(0)
    tr, _ = gen.brownian_single_state(2, Fr.T, Ds=[3, 1], generator=g)
    v = Fr.trajs_to_vid_framerates(tr.permute(1, 0, 2) / Fr.traj_div_factor, generator=g)
    assert v.shape == (2, 6, 60, 13, 13) and float(v[:, 5, 6:].abs().max()) == 0.0      # zero padding beyond 6 frames
    m, o, s = IF.getTrainingModels(addMSDModels=True)
    assert list(m) == ["im_tr", "im_ft_late_tr", "im_ft_early_tr", "im_resnet", "im_ft_resnet", "ft_mlp", "MSD_Perfect",
                       "MSD_Frame", "MSD_Localized"]
    assert m["im_ft_late_tr"].mlp_head.mlp[0].in_features == 2 * IF.embed_dim and IF.N_features == 25
    x = torch.arange(2 * 3 * 4 * 4.).reshape(2, 3, 4, 4)
    r = IF.generate_rotated_sequences(x)
    assert torch.equal(r[2], torch.rot90(x, 2, (2, 3))) and len(r) == 4


def test_extract_particle_patches_matches_padded_slicing():
    """helpers/tracking.py against the reference's definition (helpersTracking.py:513-550): pad by half, slice a square."""
    from moleculardiffusion_mivit_amd.helpers.tracking import extract_particle_patches
    rng = np.random.default_rng(0)
    img = rng.normal(size=(5, 20, 17)).astype(np.float32)
    tracks = {7: [(0, 3.2, 4.7), (1, 0.0, 0.0), (4, 19.4, 16.5), (2, 10.5, 8.5)], "b": [(3, 9, 9)], 9: []}
    got = extract_particle_patches(img, tracks, patch_size=7)
    for tid, positions in tracks.items():
        ref = []
        for frame, y, x in positions:
            y, x = int(round(y)), int(round(x))
            padded = np.pad(img[frame], pad_width=3, mode="constant")
            ref.append(padded[y:y + 7, x:x + 7])
        assert np.array_equal(got[tid], np.array(ref)), tid
    t = extract_particle_patches(torch.as_tensor(img), {1: [(2, 5, 5)]}, patch_size=3)[1]
    assert torch.is_tensor(t) and t.shape == (1, 3, 3) and torch.equal(t[0], torch.as_tensor(img[2, 4:7, 4:7]))
