"""Boundary conditions (reference: fs/boundary_condition.py).

`BoundaryCondition` / `DyeBoundaryCondition` upload the scene (mask codes: 0 fluid, 1 wall, 2 inflow,
3 outflow) and create the device context; their three `set_*_boundary_condition` methods launch the
HIP boundary kernels, which reproduce the reference kernels' serial-order semantics exactly
(csrc/fs_kernels.h "op lists").

The six scenes of `get_boundary_condition` are restated as NumPy drawing programs on a small `_Canvas`
(rectangles, discs, inflow/outflow strips, painted in the reference's order so later shapes override
earlier ones).  Their u8/f32 outputs are pinned bit-for-bit against arrays captured from the
reference builders (tests/golden/scenes.npz, scene_hashes.json).
"""
import os
from pathlib import Path

import numpy as np

from . import runtime

FLUID, WALL, INFLOW, OUTFLOW = 0, 1, 2, 3


class BoundaryCondition:
    def __init__(self, bc_const, bc_mask, device=None):
        self._init_scene(bc_const, bc_mask, None, device)

    def _init_scene(self, bc_const, bc_mask, bc_dye, device):
        bc_mask = np.asarray(bc_mask)
        self._resolution = tuple(bc_mask.shape[:2])
        self.device = device if device is not None else runtime.create_device(self._resolution)
        self._host_mask = np.ascontiguousarray(bc_mask, dtype=np.uint8)
        self.device.upload_scene(self._host_mask, bc_const, bc_dye)

    def set_velocity_boundary_condition(self, vc):
        """No-slip mirror into the 2nd wall layer, inflow = const, outflow without backflow
        (fs/boundary_condition.py:16-39); in place on `vc`."""
        self.device.velocity_bc(vc)

    def set_pressure_boundary_condition(self, pc):
        """Neumann copy from the fluid neighbour (corner: mean of two), inflow copies p[i+1], outflow p = 0
        (fs/boundary_condition.py:41-65); in place on `pc`."""
        self.device.pressure_bc(pc)

    def is_wall(self, i, j):
        return bool(self._host_mask[i, j] == WALL)

    def is_fluid_domain(self, i, j):
        return bool(self._host_mask[i, j] == FLUID)

    def get_resolution(self):
        return self._resolution

    @property
    def mask(self):
        return self._host_mask


class DyeBoundaryCondition(BoundaryCondition):
    def __init__(self, bc_const, bc_dye, bc_mask, device=None):
        self._init_scene(bc_const, bc_mask, bc_dye, device)

    def set_dye_boundary_condition(self, dye, velocity=None):
        """Inflow cells take the scene's dye colour (fs/boundary_condition.py:94-99).  velocity (new, optional): the velocity field the flow
        step just ended on - its deferred limit_field rides in this launch (runtime.DeviceBase.dye_bc)."""
        self.device.dye_bc(dye, velocity)


# ----------------------------------------------------------------------------------------------------
# scene construction (host, NumPy) - restates fs/boundary_condition.py:115-524
# ----------------------------------------------------------------------------------------------------
class _Canvas:
    """(2*res) x res drawing surface holding the three scene arrays (fs/boundary_condition.py:115-122)."""

    def __init__(self, resolution):
        self.res = int(resolution)
        self.X, self.Y = 2 * self.res, self.res
        self.vel = np.zeros((self.X, self.Y, 2), np.float32)
        self.mask = np.zeros((self.X, self.Y), np.uint8)
        self.dye = np.zeros((self.X, self.Y, 3), np.float32)

    def strip(self, xs, ys, code, velocity=(0.0, 0.0)):
        """Mark columns `xs`, rows `ys` as inflow / outflow with a prescribed velocity; dye untouched."""
        self.vel[xs, ys] = np.array(velocity)
        self.mask[xs, ys] = code

    def box(self, lo, hi):
        """Solid rectangle [lo, hi) (fs/boundary_condition.py:157-168)."""
        sl = (slice(lo[0], hi[0]), slice(lo[1], hi[1]))
        self.vel[sl] = 0.0
        self.mask[sl] = WALL
        self.dye[sl] = 0.0

    def disc(self, center, radius):
        """Solid disc: cells whose centre (i+.5, j+.5) is strictly closer than `radius` to `center`, searched
        inside the reference's rounded bounding box (fs/boundary_condition.py:137-154)."""
        c0, c1 = float(center[0]), float(center[1])
        lo = np.round(np.maximum(np.array([c0, c1]) - radius, 0)).astype(np.int32)
        hi0 = round(min(center[0] + radius, self.X))
        hi1 = round(min(center[1] + radius, self.Y))
        if hi0 <= lo[0] or hi1 <= lo[1]:
            return
        dx = (np.arange(lo[0], hi0) + 0.5) - c0
        dy = (np.arange(lo[1], hi1) + 0.5) - c1
        inside = np.sqrt((dx * dx)[:, None] + (dy * dy)[None, :]) < radius
        sl = (slice(lo[0], hi0), slice(lo[1], hi1))
        self.vel[sl][inside] = 0.0
        self.mask[sl][inside] = WALL
        self.dye[sl][inside] = 0.0

    def stencil(self, solid):
        """Solid cells from a boolean (X, Y) image."""
        self.vel[solid] = 0.0
        self.mask[solid] = WALL
        self.dye[solid] = 0.0

    def floor_and_ceiling(self):
        self.box((0, 0), (self.X, 2))
        self.box((0, self.Y - 2), (self.X, self.Y))

    def finish(self, enable_dye):
        if enable_dye:
            return DyeBoundaryCondition(self.vel, self.dye, self.mask)
        return BoundaryCondition(self.vel, self.mask)


def create_color_map(color_list, n_samples):
    """Piecewise-linear colour ramp through `color_list`, sampled at n points (fs/boundary_condition.py:125-134)."""
    knots = np.vstack(color_list)
    t = np.linspace(0.0, 1.0, knots.shape[0], endpoint=True)
    s = np.linspace(0.0, 1.0, n_samples, endpoint=True)
    return np.stack([np.interp(s, t, knots[:, ch]) for ch in range(3)], axis=1)


_YELLOW, _BLUE, _RED, _CYAN = (np.array(c) for c in ([1.1, 1.1, 0.2], [0.2, 0.2, 1.1], [1.1, 0.2, 0.2], [0.2, 1.1, 1.1]))


def _rainbow_inflow(cv, repeats):
    """Full-height inflow through columns 0-1 with a cyan-red-blue-yellow dye ramp (scenes 1, 3, 6)."""
    cv.strip(slice(0, 2), slice(None), INFLOW, (1.0, 0.0))
    ramp = create_color_map([_CYAN, _RED, _BLUE, _YELLOW] * repeats, cv.Y)
    cv.dye[:2, :] = ramp[None, :, :]


def _scene1(cv):
    """Cylinder in a channel (fs/boundary_condition.py:222-266)."""
    _rainbow_inflow(cv, 3)
    cv.strip(-1, slice(None), OUTFLOW)
    cv.floor_and_ceiling()
    cv.disc((cv.X // 4, cv.Y // 2), cv.Y // 18)


def _scene2(cv):
    """Serpentine: four alternating baffles, inflow/outflow through the middle third (fs/boundary_condition.py:269-320)."""
    X, Y = cv.X, cv.Y
    cv.strip(slice(0, 2), slice(None), INFLOW, (1.0, 0.0))
    cv.dye[:2, :] = np.array([0.2, 0.2, 1.2])
    band = Y // 10
    for j in range(0, Y, band):
        cv.dye[:2, j:j + band // 2] = np.array([1.2, 1.2, 0.2])
    cv.box((0, 0), (2, Y // 3))
    cv.box((0, 2 * Y // 3), (2, Y))
    cv.box((X - 2, 0), (X, Y))
    cv.floor_and_ceiling()
    xs, ym, half = X // 5, Y // 2, Y // 32
    cv.box((xs - half, ym), (xs + half, Y))
    cv.box((2 * xs - half, 0), (2 * xs + half, ym))
    cv.box((3 * xs - half, ym), (3 * xs + half, Y))
    cv.box((4 * xs - half, 0), (4 * xs + half, ym))
    third = Y // 3
    cv.strip(slice(X - 2, X), slice(third, 2 * third), OUTFLOW)


def _scene3(cv):
    """Random cylinders (legacy global RNG, seed 123) in a channel (fs/boundary_condition.py:323-370)."""
    _rainbow_inflow(cv, 1)
    cv.strip(-1, slice(None), OUTFLOW)
    cv.floor_and_ceiling()
    np.random.seed(123)  # noqa: NPY002 - the reference's scene IS this legacy stream
    centres = np.random.uniform(0, cv.X, (100, 2))  # noqa: NPY002
    centres = centres[centres[:, 1] < cv.Y]
    radius = 16 * (cv.Y / 500)
    for c in centres:
        cv.disc(c, radius)


def _scene4(cv):
    """Closed box with two inflow slots on the left and one outflow slot on the right (fs/boundary_condition.py:373-419)."""
    X, Y = cv.X, cv.Y
    cv.box((0, 0), (2, Y))
    cv.box((X - 2, 0), (X, Y))
    cv.floor_and_ceiling()
    ramp = create_color_map([_CYAN, _RED, _BLUE, _YELLOW], Y // 4 - 2)
    upper, lower = slice(3 * Y // 4, Y - 2), slice(2, Y // 4)
    for slot in (upper, lower):
        cv.dye[:2, slot] = ramp[None, :, :]
    for slot in (upper, lower):
        cv.strip(slice(0, 2), slot, INFLOW, (1.0, 0.0))
    cv.strip(slice(X - 2, X), slice(3 * Y // 8, 5 * Y // 8), OUTFLOW)


def _scene5(cv):
    """Two inlets around a block, a slotted mid wall and a staggered array of square posts (fs/boundary_condition.py:422-479)."""
    X, Y = cv.X, cv.Y
    for slot, colour in ((slice(2, Y // 3), [1.2, 0.2, 0.2]), (slice(2 * Y // 3, Y - 2), [0.2, 1.2, 1.2])):
        cv.strip(slice(0, 2), slot, INFLOW, (1.0, 0.0))
        cv.dye[:2, slot] = np.array(colour)
    cv.strip(slice(X - 2, X), slice(None), OUTFLOW)
    cv.floor_and_ceiling()
    half = X // 64
    cv.box((0, Y // 5), (11 * X // 30, 4 * Y // 5))
    cv.box((X // 2 - half, 0), (X // 2 + half, 2 * Y // 5))
    cv.box((X // 2 - half, 3 * Y // 5), (X // 2 + half, Y))
    pitch, post = Y // 6, np.array([Y, Y]) // 25
    for column, shifted in zip((7, 8, 9, 10, 11), (0, 1, 0, 1, 0)):
        for n in range(1, 6 + shifted):
            centre = np.array([column * X // 12, n * pitch - shifted * Y // 12])
            cv.box(centre - post, centre + post)


def _find_obstacle_image(name="dragon.png"):
    roots = [os.environ.get("FS_ASSET_DIR"), Path(__file__).resolve().parents[1], Path(__file__).resolve().parents[2], Path.cwd()]
    for root in roots:
        if root:
            p = Path(root) / "images" / "bc_mask" / name
            if p.exists():
                return p
    raise FileNotFoundError(
        f"scene 6 needs the obstacle image images/bc_mask/{name} (an asset of the reference repository); "
        "set FS_ASSET_DIR to a directory that contains it")


def obstacle_from_image(cv, filepath):
    """Dark pixels (< 200 of 255) of a grayscale image, fitted into the domain keeping the aspect ratio and
    centred horizontally, become wall (fs/boundary_condition.py:171-198).  Pillow-version sensitive."""
    from PIL import Image

    img = Image.open(filepath).convert("L")
    sx, sy = cv.X / img.width, cv.Y / img.height
    size = (cv.X, round(img.height * sx)) if sx < sy else (round(img.width * sy), cv.Y)
    img = img.resize(size)
    sheet = Image.new(img.mode, (cv.X, cv.Y), 255)
    sheet.paste(img, ((cv.X - img.width) // 2, 0))
    cv.stencil(np.flip(np.array(sheet).T, axis=1) < 200)


def _scene6(cv):
    """Image-defined obstacle in a channel (fs/boundary_condition.py:482-524)."""
    _rainbow_inflow(cv, 1)
    cv.strip(-1, slice(None), OUTFLOW)
    cv.floor_and_ceiling()
    obstacle_from_image(cv, _find_obstacle_image())


_SCENES = {1: _scene1, 2: _scene2, 3: _scene3, 4: _scene4, 5: _scene5, 6: _scene6}


def create_scene_arrays(num, resolution):
    """(bc_const (X, Y, 2) f32, bc_mask (X, Y) u8, bc_dye (X, Y, 3) f32) of scene `num`, host only."""
    if num not in _SCENES:
        raise NotImplementedError
    cv = _Canvas(resolution)
    _SCENES[num](cv)
    return cv.vel, cv.mask, cv.dye


def get_boundary_condition(num, resolution, *, enable_dye):
    """fs/boundary_condition.py:201-219."""
    if num not in _SCENES:
        raise NotImplementedError
    cv = _Canvas(resolution)
    _SCENES[num](cv)
    return cv.finish(enable_dye)


def _make_creator(num):
    def create(resolution, *, enable_dye):
        return get_boundary_condition(num, resolution, enable_dye=enable_dye)
    create.__name__ = f"create_boundary_condition{num}"
    create.__doc__ = f"Scene {num}; see get_boundary_condition."
    return create


create_boundary_condition1, create_boundary_condition2, create_boundary_condition3 = (_make_creator(n) for n in (1, 2, 3))
create_boundary_condition4, create_boundary_condition5, create_boundary_condition6 = (_make_creator(n) for n in (4, 5, 6))
