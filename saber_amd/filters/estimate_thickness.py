"""Presence scores along z from the mask decoder's object-score logits (reference: saber/filters/estimate_thickness.py:6-112, called by
SAM2Adapter.segment_volume, adapters/sam2/predictor.py:325-346).  Per tracked object the logits over the frames are clipped at zero,
shifted by the mean of frames [-15:-5], clipped again, and two bounded bump models are fitted with scipy's curve_fit - a clipped
parabola and a Gaussian; the one with the better R^2 gives every frame's presence score.  Volume-level host glue outside the per-frame
model, numpy / scipy like the reference."""
import numpy as np


def quadratic(x, a, b, c, d):
    return d * np.maximum(a * (x - b) ** 2 + c, 0)


def gaussian(x, a, b, c):
    with np.errstate(over="ignore"):
        return a * np.exp(-(x - b) ** 2 / (2 * c ** 2))


def calculate_r2_score(data, func, fit_params):
    x = np.arange(len(data))
    y_fit = func(x, *fit_params)
    ss_res = np.sum((data - y_fit) ** 2)
    ss_tot = np.sum((data - np.mean(data)) ** 2)
    return 0 if ss_tot == 0 else 1 - ss_res / ss_tot


def preprocess(data: np.ndarray) -> np.ndarray:
    data = np.maximum(data, 0)
    data -= np.mean(data[-15:-5])
    return np.maximum(data, 0)


def fit_quadratic(x, data):
    from scipy.optimize import curve_fit
    n = data.shape[0]
    popt, _ = curve_fit(quadratic, x, data, p0=[-1e-3, np.argmax(data[1:-1]), 1, np.max(data) / 2], bounds=([-np.inf, 0, 0, 0], [0, n, 10, 10]))
    return popt, calculate_r2_score(data, quadratic, popt)


def fit_gaussian(x, data):
    from scipy.optimize import curve_fit
    n = data.shape[0]
    popt, _ = curve_fit(gaussian, x, data, p0=[np.max(data), np.argmax(data[1:-1]), 3e-1], bounds=((0, 0, 0), (np.inf, n, n * 0.25 / 2.355)))
    return popt, calculate_r2_score(data, gaussian, popt)


def fit_organelle_boundaries(frame_scores: np.ndarray, plot: bool = False) -> np.ndarray:
    """frame_scores (n_frames, n_masks) -> presence score per frame and mask (same shape)"""
    n_frames, n_masks = frame_scores.shape
    out = np.zeros((n_frames, n_masks))
    for ii in range(n_masks):
        data = preprocess(frame_scores[:, ii].copy())
        x = np.arange(len(data), dtype=np.float32)
        try:
            p1, r1 = fit_quadratic(x, data)
        except Exception as e:            # curve_fit gives up on degenerate input: the reference prints and scores the model 0
            print(f"Error fitting Quadratic mask {ii}: {e}")
            r1 = 0
        try:
            p2, r2 = fit_gaussian(x, data)
        except Exception as e:
            print(f"Error fitting Gaussian mask {ii}: {e}")
            r2 = 0
        if r1 == 0 and r2 == 0:
            out[:, ii] = np.zeros(data.shape[0])
        elif r1 > r2:
            out[:, ii] = quadratic(x, *p1)
        else:
            out[:, ii] = gaussian(x, *p2)
    return out
