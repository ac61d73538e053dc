"""Headless CLI (2d-fluid-simulator_amd/main.py): the reference's flag set (main.py:11-51) and the npz dump format
(main.py:129-132); exact restart from the full-state checkpoint (new)."""
import importlib.util
import os

import numpy as np
import pytest
from conftest import REPO


def _cli():
    spec = importlib.util.spec_from_file_location("fs_cli_main", os.path.join(REPO, "2d-fluid-simulator_amd", "main.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_flags_and_defaults_match_the_reference():
    a = _cli().build_parser().parse_args([])
    assert (a.boundary_condition, a.reynolds_num, a.resolution, a.time_step) == (1, 1000000.0, 400, 0.0)
    assert (a.visualization, a.vorticity_confinement, a.advection_scheme, a.no_dye, a.cpu) == (0, 5.0, "cip", False, False)
    b = _cli().build_parser().parse_args("-bc 3 -re 100000000 -res 800 -vc 10 -scheme kk -no_dye -vis 2 -dt 0.0005".split())
    assert (b.boundary_condition, b.reynolds_num, b.resolution, b.vorticity_confinement) == (3, 1e8, 800, 10.0)
    assert (b.advection_scheme, b.no_dye, b.visualization, b.time_step) == ("kk", True, 2, 0.0005)
    with pytest.raises(SystemExit):
        _cli().build_parser().parse_args(["-bc", "7"])
    with pytest.raises(SystemExit):
        _cli().build_parser().parse_args(["-scheme", "weno"])


@pytest.mark.gpu
def test_dump_format_and_exact_restart(tmp_path, hip_lib):
    cli = _cli()
    common = "-bc 2 -res 64 -vc 5".split()
    a, b = tmp_path / "a", tmp_path / "b"
    cli.main(common + ["--steps", "6", "--dump-every", "6", "--out", str(a), "--frame-every", "3"])
    cli.main(common + ["--steps", "3", "--out", str(b), "--save-state", str(b / "ck.npz")]) if b.mkdir() is None else None
    cli.main(common + ["--steps", "3", "--out", str(b), "--load-state", str(b / "ck.npz"), "--dump-every", "3"])
    full = np.load(a / "step_000006.npz")
    resumed = np.load(b / "step_000006.npz")
    assert sorted(full.files) == ["dye", "p", "v"]                       # keys of the reference's `d` dump
    assert full["v"].shape == (128, 64, 2) and full["p"].shape == (128, 64) and full["dye"].shape == (128, 64, 3)
    for k in full.files:
        assert np.array_equal(full[k], resumed[k]), k                    # checkpoint restart is exact
    assert (a / "000000.png").exists() and (a / "000003.png").exists()


@pytest.mark.gpu
def test_visualisation_buffers(hip_lib):
    import fs
    fs.runtime.init(gpu=0)
    sim = fs.DyeFluidSimulator.create(1, 32, 0.05 / 32, 1 / 32, 1e6, 5.0, "cip")
    for _ in range(3):
        sim.step()
    wall = sim._solver._bc.mask == 1
    for field in (sim.get_norm_field(), sim.get_pressure_field(), sim.get_vorticity_field(), sim.get_dye_field()):
        img = field.to_numpy()
        assert img.shape == (64, 32, 3) and np.isfinite(img).all()
        assert np.allclose(img[wall], [0.5, 0.7, 0.5])                   # wall colour, fluid_simulator.py:17
    sim._solver._bc.device.close()


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
def test_dump_contents_equal_the_reference_trajectory(tmp_path, hip_lib, graph):
    """the `d`-key dump (main.py:129-132) of `-bc 2 -res 32 -no_dye` (defaults: cip, vc 5, Re 1e6, dt 0.05/res)
    holds exactly the fields the reference's own run produces (tests/golden/traj_bc2_cip_vc5.npz), with and without --graph."""
    from conftest import golden
    g = golden("traj_bc2_cip_vc5.npz")
    cli = _cli()
    cli.main("-bc 2 -res 32 -no_dye --steps 10 --dump-every 5".split() + ["--out", str(tmp_path)] + (["--graph"] if graph else []))
    for step in (5, 10):
        z = np.load(tmp_path / f"step_{step:06}.npz")
        assert sorted(z.files) == ["p", "v"]
        for k in ("v", "p"):
            assert z[k].dtype == g[f"step{step}.{k}"].dtype and np.array_equal(z[k], g[f"step{step}.{k}"]), (step, k)


@pytest.mark.gpu
def test_dye_dump_equals_the_reference_trajectory(tmp_path, hip_lib):
    """The reference's default mode (dye on): `-bc 5 -res 32 -scheme kk` dump vs traj_dye_bc5_kk_vc5.npz."""
    from conftest import golden
    g = golden("traj_dye_bc5_kk_vc5.npz")
    _cli().main("-bc 5 -res 32 -scheme kk --steps 5 --dump-every 5".split() + ["--out", str(tmp_path)])
    z = np.load(tmp_path / "step_000005.npz")
    assert sorted(z.files) == ["dye", "p", "v"]
    for k in z.files:
        assert np.array_equal(z[k], g[f"step5.{k}"]), k


@pytest.mark.gpu
def test_checkpoint_name_without_extension(tmp_path, hip_lib):
    """`--save-state ckpt` writes ckpt.npz (np.savez appends the suffix); `--load-state ckpt` must find it."""
    cli = _cli()
    cli.main("-bc 1 -res 32 -no_dye --steps 2".split() + ["--out", str(tmp_path), "--save-state", str(tmp_path / "ckpt")])
    assert (tmp_path / "ckpt.npz").exists()
    cli.main("-bc 1 -res 32 -no_dye --steps 1".split() + ["--out", str(tmp_path), "--load-state", str(tmp_path / "ckpt")])


def test_scene6_without_asset_is_an_argparse_error(monkeypatch, tmp_path, capsys):
    """`-bc 6` is an accepted choice but needs the reference's dragon.png: a clear usage error, not a traceback."""
    from fs import boundary_condition as B
    monkeypatch.setattr(B, "_find_obstacle_image", lambda name="dragon.png": (_ for _ in ()).throw(
        FileNotFoundError("scene 6 needs the obstacle image images/bc_mask/dragon.png; set FS_ASSET_DIR")))
    with pytest.raises(SystemExit) as e:
        _cli().main(["-bc", "6", "-res", "32", "--steps", "1", "--out", str(tmp_path)])
    assert e.value.code == 2
    assert "FS_ASSET_DIR" in capsys.readouterr().err
