"""The product's NumPy scene builders (fs/boundary_condition.py of the package) against arrays and hashes
captured from the reference builders (reference fs/boundary_condition.py:222-524).  Bit-exact u8 / f32."""
import hashlib
import json
import os

import numpy as np
import pytest
from conftest import GOLDEN, golden


def _sha(a):
    return hashlib.sha256(a.tobytes()).hexdigest()[:16]


@pytest.mark.parametrize("res", [16, 32, 48])
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5])
def test_scene_arrays_bit_exact(n, res):
    from fs.boundary_condition import create_scene_arrays
    g = golden("scenes.npz")
    const, mask, dye = create_scene_arrays(n, res)
    for key, a in (("bc_const", const), ("bc_mask", mask), ("bc_dye", dye)):
        e = g[f"bc{n}_res{res}_{key}"]
        assert a.dtype == e.dtype and np.array_equal(a, e), (n, res, key)


def test_scene_hashes_at_reference_sizes():
    """Includes BASELINE.json's configs: bc1@200, bc2@1600, bc5@4096, bc3@4096 (bc2@8192: next test)."""
    from fs.boundary_condition import create_scene_arrays
    H = json.load(open(os.path.join(GOLDEN, "scene_hashes.json")))["scenes"]
    checked = 0
    for key, e in sorted(H.items()):
        n, res = int(key[2]), int(key.split("res")[1])
        if n == 6 or res > 4096:
            continue
        const, mask, dye = create_scene_arrays(n, res)
        assert list(mask.shape) == e["shape"]
        assert [int((mask == c).sum()) for c in range(4)] == e["counts"], key
        assert (_sha(mask), _sha(const), _sha(dye)) == (e["sha_mask"], e["sha_bc_const"], e["sha_bc_dye"]), key
        checked += 1
    assert checked >= 8


def test_scene_hash_bc2_res8192():
    """BASELINE.json configs[3]'s scene (16384 x 8192 cells, reference fs/boundary_condition.py:269-320): counts and SHA-256 of
    mask / bc_const / bc_dye as captured from the reference builder.  About 3 GB of host arrays for half a minute."""
    from fs.boundary_condition import create_scene_arrays
    e = json.load(open(os.path.join(GOLDEN, "scene_hashes.json")))["scenes"]["bc2_res8192"]
    const, mask, dye = create_scene_arrays(2, 8192)
    assert list(mask.shape) == e["shape"] == [16384, 8192]
    assert [int((mask == c).sum()) for c in range(4)] == e["counts"]
    assert (_sha(mask), _sha(const), _sha(dye)) == (e["sha_mask"], e["sha_bc_const"], e["sha_bc_dye"])


REF_ASSETS = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_ASSETS, "images", "bc_mask", "dragon.png")),
                    reason="scene 6 needs the reference's dragon.png (present in the build container only; not copied)")
@pytest.mark.parametrize("res", [16, 32, 48])
def test_scene6_from_the_reference_asset(res, monkeypatch):
    """the image-obstacle scene (reference fs/boundary_condition.py:171-198, 482-524) built by the product's
    scene builder from the reference's own asset (looked up through FS_ASSET_DIR, never copied) equals the arrays the
    reference builder produced (tests/golden/scenes.npz), bit for bit."""
    from fs.boundary_condition import create_scene_arrays
    monkeypatch.setenv("FS_ASSET_DIR", REF_ASSETS)
    g = golden("scenes.npz")
    const, mask, dye = create_scene_arrays(6, res)
    for key, a in (("bc_const", const), ("bc_mask", mask), ("bc_dye", dye)):
        e = g[f"bc6_res{res}_{key}"]
        assert a.dtype == e.dtype and np.array_equal(a, e), (res, key)
    assert 0 < int((mask[2:-2, 2:-2] == 1).sum()) < mask[2:-2, 2:-2].size      # the dragon is there, and it is not everything


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_ASSETS, "images", "bc_mask", "dragon.png")), reason="needs the reference asset")
def test_scene6_hash_at_reference_size(monkeypatch):
    from fs.boundary_condition import create_scene_arrays
    monkeypatch.setenv("FS_ASSET_DIR", REF_ASSETS)
    e = json.load(open(os.path.join(GOLDEN, "scene_hashes.json")))["scenes"]["bc6_res400"]
    const, mask, dye = create_scene_arrays(6, 400)
    assert [int((mask == c).sum()) for c in range(4)] == e["counts"]
    assert (_sha(mask), _sha(const), _sha(dye)) == (e["sha_mask"], e["sha_bc_const"], e["sha_bc_dye"])


def test_scene_invariants():
    """Facts every reference scene satisfies: ring never fluid, rows 0,1,Y-2,Y-1 all wall,
    inflow in columns 0-1 only, outflow in the last one or two columns."""
    from fs.boundary_condition import create_scene_arrays
    for n in (1, 2, 3, 4, 5):
        _, m, _ = create_scene_arrays(n, 64)
        X, Y = m.shape
        assert (m[:, [0, 1, Y - 2, Y - 1]] == 1).all()
        assert (m[[0, X - 1], :] != 0).all()
        assert not (m[2:, :] == 2).any()
        assert not (m[:X - 2, :] == 3).any()


def test_unknown_scene_raises():
    from fs.boundary_condition import create_scene_arrays
    with pytest.raises(NotImplementedError):
        create_scene_arrays(7, 16)


def test_scene6_needs_the_reference_asset(monkeypatch, tmp_path):
    from fs import boundary_condition as B
    monkeypatch.setenv("FS_ASSET_DIR", str(tmp_path))
    monkeypatch.chdir(tmp_path)
    try:
        B._find_obstacle_image()
    except FileNotFoundError as e:
        assert "dragon.png" in str(e)
