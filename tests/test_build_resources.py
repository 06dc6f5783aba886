"""The built code object must keep the step kernels at four workgroups per CU: <= 128 VGPRs per lane (512 / 4 waves per
SIMD) and an SGPR count that still admits four 256-thread workgroups.  Launch bounds only bind KERNELS: a device
function the compiler decides to outline (or one marked noinline) allocates registers on its own and silently drags
the kernels that call it to occupancy 2 -- this test reads the numbers from the library that ships."""
import os
import re
import shutil
import subprocess

import pytest

from rlfluidcontrol_amd import build

LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "clang-offload-bundler")) or shutil.which("objcopy") is None,
                    reason="ROCm LLVM tools not available")
def test_step_kernels_keep_four_workgroups_per_cu(tmp_path):
    lib = build.build()
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    kernels = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        kernels[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
                         for k in ("vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size")}
        kernels[name]["agpr_count"] = int(re.match(r"\s*(\d+)", blk).group(1))
    step = {n: v for n, v in kernels.items() if "k_agent_step" in n or "k_sim_step" in n}
    # both kernels: CPL = 0 (reference kernels) + CPL = 1..4 x {shipped, diagnostics compiled in}
    assert len(step) == 18
    shipped = {n: v for n, v in step.items() if "Lb0E" in n}
    assert len(shipped) == 8
    for n, v in shipped.items():
        assert v["private_segment_fixed_size"] <= 640, (n, v)  # spill budget (bytes per lane) of any step kernel
        if "k_agent_step" in n:  # the bench / training kernel: no scratch growth against the diagnostic instantiation
            d = step[n.replace("Lb0E", "Lb1E")]
            assert v["private_segment_fixed_size"] <= d["private_segment_fixed_size"], (n, v, d)
    for n, v in step.items():
        assert v["vgpr_count"] + v["agpr_count"] <= 128, (n, v)
        # MI355X_MICROARCH.md: 256-thread blocks per CU <= floor(800 / (ceil(sgpr/16)*16 + 16))
        assert 800 // (-(-v["sgpr_count"] // 16) * 16 + 16) >= 4, (n, v)
        if "ILi0E" not in n:  # optimised kernels: static LDS small, the dynamic part (tiles / rings) gets the rest of 40 KB
            assert v["group_segment_fixed_size"] <= 3 * 1024, (n, v)
