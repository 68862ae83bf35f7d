"""The 25 hand-crafted trajectory descriptors of the ImagesFeatures experiment (reference helpers/helpersFeatures.py:7-34
names, :448-519 compute_diffusion_features and the helpers it calls :36-444): the ``features[B, 25]`` tensor that
GeneralTransformer's early / late fusion consumes (models.py:341-345,356-359).

CPU pre-processing on sub-pixel-averaged trajectories of nFrames = 30 points, outside the GPU hot path (SURVEY section 8 row
f4).  Restated with vectorised numpy (pairwise distances, all lags at once) instead of the reference's Python loops; the
two library calls the reference makes are made the same way -- scipy.optimize.curve_fit (bounded 'trf' fit of
MSD = 4 D t^alpha + offset) and scipy.spatial.ConvexHull.  Pinned against the reference's outputs by
tests/golden/features.npz (tests/test_features.py)."""
import numpy as np

feature_names = [
    "alpha", "diffusion_coefficient", "r_squared", "efficiency_log", "efficiency", "fractal_dimension", "gaussianity",
    "kurtosis", "msd_ratio", "trappedness", "trajectory_length", "mean_step_length", "mean_msd", "mean_dot_product",
    "fraction_same_direction", "fraction_positive_direction", "total_distance", "min_step", "max_step", "step_range",
    "avg_velocity", "step_cv", "fraction_small_steps", "fraction_large_steps", "convex_hull_area"]
N_features = len(feature_names)


def _lag_moments(p, nlags):
    """mean |p[j+lag] - p[j]|^2 and mean (dx^4 + dy^4) for lag = 1 .. nlags (reference msd :102-132, gaussianity :250-284)."""
    m2, m4 = np.empty(nlags), np.empty(nlags)
    for lag in range(1, nlags + 1):
        d = p[lag:] - p[:-lag]
        m2[lag - 1] = np.mean((d ** 2).sum(1))
        m4[lag - 1] = np.mean((d ** 4).sum(1))
    return m2, m4


def _fit_power_law(msds, dt, dim=2):
    """Bounded fit of MSD(t) = 2 dim D t^alpha + offset (reference fit_diffusion_scaling :135-191)."""
    from scipy.optimize import curve_fit
    t = np.arange(1, len(msds) + 1) * dt

    def power_law(tt, D, alpha, offset):
        return 2 * dim * D * tt ** alpha + offset
    try:
        params, _ = curve_fit(power_law, t, msds, p0=[msds[0] / (4 * dt), 1, 0.001],
                              bounds=([0.00001, 0.00001, 0], [np.inf, 10, np.inf]), method="trf", maxfev=10000)
        res = msds - power_law(t, *params)
        return params, 1 - np.sum(res ** 2) / np.sum((msds - np.mean(msds)) ** 2)
    except (RuntimeError, ValueError):
        return (0, 0), 0


def compute_diffusion_features(trajectory, dt=1.0):
    """(N, 2) positions -> the 25 descriptors in `feature_names` order (NaN vector for fewer than 3 points)."""
    p = np.asarray(trajectory, dtype=np.float64)
    n = len(p)
    if n < 3:
        return np.array([np.nan] * N_features)
    x, y = p[:, 0], p[:, 1]
    nl = (int(n * 0.5) if n > 20 else n) - 1                       # lags 1 .. nl
    msd, r4 = _lag_moments(p, nl)
    diff = p[:, None, :] - p[None, :, :]
    max_sq = float((diff ** 2).sum(-1).max())                       # largest squared pair distance (get_max_dist :70-99)
    steps = p[1:] - p[:-1]
    sl = np.sqrt((steps ** 2).sum(1))
    dots = (steps[:-1] * steps[1:]).sum(1) if len(steps) > 1 else np.array([0])
    params, r2 = _fit_power_law(msd, dt)
    D, alpha = params[0], params[1]
    # efficiency (:194-218)
    bottom = float((sl ** 2).sum())
    top = float(((p[-1] - p[0]) ** 2).sum())
    if bottom == 0:
        eff_log, eff = -np.inf, 0
    else:
        eff = top / ((n - 1) * bottom)
        eff_log = np.log(eff)
    # fractal dimension, Katz & George (:221-247)
    total = float(sl.sum())
    fractal = 1 if total == 0 else np.log(n) / (np.log(n) + np.log(np.sqrt(max_sq) / total))
    # gaussianity (:250-284): mean over lags of <r^4> / (2 <r^2>^2), lags with zero MSD skipped
    ok = msd > 0
    gauss = np.mean(r4[ok] / (2 * msd[ok] ** 2)) if ok.any() else np.nan
    # kurtosis of the projection on the dominant axis (:287-324), Pearson definition
    try:
        val, vec = np.linalg.eig(np.cov(x, y))
        dom = vec[:, np.argsort(val)][:, -1]
        proj = p @ dom
        c = proj - proj.mean()
        kurt = np.mean(c ** 4) / np.mean(c ** 2) ** 2
    except Exception:
        kurt = np.nan
    msd_ratio = np.mean(msd[:-1] / msd[1:] - np.arange(1, nl) / np.arange(2, nl + 1)) if nl >= 2 else np.nan
    r0 = np.sqrt(max_sq) / 2
    trapped = 0 if (r0 == 0 or D == 0) else 1 - np.exp(0.2045 - 0.25117 * (D * n) / r0 ** 2)
    try:
        from scipy.spatial import ConvexHull
        hull_area = ConvexHull(p).volume
    except Exception:
        hull_area = 0
    mean_sl = np.nanmean(sl)
    return np.array([
        alpha, D, r2, eff_log, eff, fractal, gauss, kurt, msd_ratio, trapped, n, mean_sl, np.nanmean(msd),
        np.nanmean(dots) if len(dots) > 0 else np.nan,
        np.nanmean(np.sign(dots[1:]) == np.sign(dots[:-1])) if len(dots) > 1 else np.nan,
        np.nanmean(np.sign(dots) > 0) if len(dots) > 0 else np.nan,
        np.nansum(sl), np.nanmin(sl), np.nanmax(sl), np.nanmax(sl) - np.nanmin(sl), np.nansum(sl) / n,
        np.nanstd(sl, ddof=1) / mean_sl if mean_sl > 0 and len(sl) > 1 else np.nan,
        np.nansum(sl < 0.1) / len(sl), np.nansum(sl > 0.4) / len(sl), hull_area])


def average_trajectories_frames(trajectories, nPosFrame):
    """(N, T, 2) -> (N, T // nPosFrame, 2): mean position of every frame's sub-steps (helpersGeneration.py:48-68)."""
    t = np.asarray(trajectories)
    n, steps, d = t.shape
    nf = steps // nPosFrame
    return t[:, :nf * nPosFrame].reshape(n, nf, nPosFrame, d).mean(axis=2)


def compute_features_for_trajectories(trajectories, nPosPerFrame, dt=1.0, localization_uncertainty=(0, 0), rng=None):
    """What create_video_and_feature_pairs does on the trajectory side (helpersGeneration.py:703-718): frame-averaged
    positions (+ the same with Gaussian localisation error) and the [N, 25] feature matrix of the averaged ones."""
    avg = average_trajectories_frames(trajectories, nPosPerFrame)
    rng = rng or np.random.default_rng()
    noisy = avg + rng.normal(localization_uncertainty[0], localization_uncertainty[1], size=avg.shape)
    feats = np.stack([compute_diffusion_features(a, dt=dt) for a in avg]) if len(avg) else np.zeros((0, N_features))
    return feats, avg, noisy
