"""Colour maps for the reference's viewer (fs/visualization.py:8-22), as host-side NumPy on downloaded
fields.  GUI-side code: not on the step() hot path, no GPU kernel spent on it."""
import numpy as np


def visualize_norm(v):
    c = np.sqrt(v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1])
    return np.stack([c, c, c], axis=-1)


def visualize_pressure(p):
    z = np.zeros_like(p)
    return np.stack([np.maximum(p, 0.0), z, np.maximum(-p, 0.0)], axis=-1)


def _central(f, axis, dx):
    g = np.take(f, np.clip(np.arange(f.shape[axis]) + 1, 0, f.shape[axis] - 1), axis=axis) \
        - np.take(f, np.clip(np.arange(f.shape[axis]) - 1, 0, f.shape[axis] - 1), axis=axis)
    return (np.float32(0.5) * g / np.float32(dx)).astype(f.dtype)


def visualize_vorticity(v, dx):
    w = _central(v[..., 1], 0, dx) - _central(v[..., 0], 1, dx)
    z = np.zeros_like(w)
    return np.stack([np.maximum(w, 0.0), z, np.maximum(-w, 0.0)], axis=-1)
