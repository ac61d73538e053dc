"""What the compiler made of the hot kernels, read from the code objects inside the built libfs_hip.so (no GPU needed): the kernels whose
speed rests on a register budget must not have slipped into scratch - the allocator is sensitive to the shape of the source (round 5: a
re-arranged division in cip_point gave the one-launch fs_cip_step body 20 bytes of scratch at power-of-two dx, which every workgroup of
the launch pays for) - and must keep the occupancy their launch geometry was chosen for."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "2d-fluid-simulator_amd", "csrc", "libfs_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not os.path.exists(LIB):
        pytest.skip("libfs_hip.so not built")
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm llvm tools not found")
    d = str(tmp_path_factory.mktemp("co"))
    fat = os.path.join(d, "fat.bin")
    subprocess.run([tools[0], "--dump-section", ".hip_fatbin=" + fat, LIB, os.path.join(d, "copy.so")], check=True)
    data = open(fat, "rb").read()
    offs = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]      # one bundle per translation unit
    assert offs
    out = {}
    for n, o in enumerate(offs):
        b, co = os.path.join(d, f"b{n}.bin"), os.path.join(d, f"co{n}.o")
        open(b, "wb").write(data[o:offs[n + 1] if n + 1 < len(offs) else len(data)])
        subprocess.run([tools[1], "--unbundle", "--type=o", "--input", b, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output", co], check=True)
        notes = subprocess.run([tools[2], "--notes", co], capture_output=True, text=True, check=True).stdout
        cur = {}
        for line in notes.splitlines():
            m = re.search(r"\.(name|private_segment_fixed_size|vgpr_count|group_segment_fixed_size):\s+(\S+)", line)
            if m:
                cur[m.group(1)] = m.group(2)
            if "name" in cur and "private_segment_fixed_size" in cur and "vgpr_count" in cur:
                out[cur["name"]] = {"scratch": int(cur["private_segment_fixed_size"]), "vgprs": int(cur["vgpr_count"]), "lds": int(cur.get("group_segment_fixed_size", 0))}
                cur = {}
    shutil.rmtree(d, ignore_errors=True)
    return out


def _pick(kernels, pattern):
    got = {k: v for k, v in kernels.items() if re.search(pattern, k)}
    assert got, pattern
    return got


def test_the_library_holds_the_gfx950_code_objects(kernels):
    assert len(kernels) > 200 and any("k_cip_step_all" in k for k in kernels)


@pytest.mark.parametrize("pattern,max_vgprs", [
    (r"14k_cip_step_allILi4ELi\dE", 128),            # fs_cip_step, one launch over every tile: 4 waves per SIMD, every division mode
    (r"16k_cip_step_plainILi4ELi\dE", 128),
    (r"16k_rbsor_pair_allILi2ELi\dELi\dEfE", 168),          # the red-black pair, one launch over both kinds of tile: 3 waves per SIMD (the all-fluid body is indifferent to 5 / 3 / 2.5)
    (r"8k_vort_nILi2ELi4ELi\dELb0EfE", 72),          # 7 - 8 waves per SIMD
    (r"12k_jacobi_ov2ILi4ELi\dE", 64),               # the graded Jacobi sweep
])
def test_hot_kernels_fit_their_register_budget_without_scratch(kernels, pattern, max_vgprs):
    for name, k in _pick(kernels, pattern).items():
        assert k["scratch"] == 0, (name, k)
        assert k["vgprs"] <= max_vgprs, (name, k)
