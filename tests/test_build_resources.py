"""The built code objects must keep the step kernels at four workgroups per CU: <= 128 VGPRs per lane (512 / 4 waves per
SIMD) and an SGPR count that still admits four 256-thread workgroups.  Launch bounds only bind KERNELS: a device
function the compiler decides to outline (or one marked noinline) allocates registers on its own and silently drags
the kernels that call it to occupancy 2 -- this test reads the numbers from the library that ships.  The shipped
instantiation (mode 0) must not cost more scratch or code than the one with the diagnostics compiled in (mode 1)."""
import os
import shutil

import pytest

from rlfluidcontrol_amd import build, codeobj


@pytest.mark.skipif(not os.path.exists(os.path.join(codeobj.LLVM, "clang-offload-bundler")) or shutil.which("objcopy") is None,
                    reason="ROCm LLVM tools not available")
def test_step_kernels_keep_four_workgroups_per_cu():
    kernels = codeobj.kernel_resources(build.build())
    step = {n: v for n, v in kernels.items() if "k_agent_step" in n or "k_sim_step" in n}
    # both kernels: CPL = 0 (reference kernels, diagnostic unit) + CPL = 1..4 x {shipped, diagnostics, perf mode}
    assert len(step) == 2 * (1 + 4 * 3), sorted(step)
    for n, v in step.items():
        assert v["vgpr_count"] + v["agpr_count"] <= 128, (n, v)
        # MI355X_MICROARCH.md: 256-thread blocks per CU <= floor(800 / (ceil(sgpr/16)*16 + 16))
        assert 800 // (-(-v["sgpr_count"] // 16) * 16 + 16) >= 4, (n, v)
        if "ILi0E" not in n:  # optimised kernels: static LDS small, the dynamic part (tiles / rings) gets the rest of 40 KB
            assert v["group_segment_fixed_size"] <= 3 * 1024, (n, v)
            # spill budget (bytes per lane): what is spilled sits outside the innermost loops (tools/loop_scan.py)
            # (the projection tail is a function of its own since round 4: its callee-saved registers are part of the figure;
            # the diagnostic instantiation, which nobody times, gets a little more room)
            diag = n.split("ELi")[-1].startswith("1E")
            assert v["private_segment_fixed_size"] <= ((768 if diag else 640) if "k_agent_step" in n else 1024), (n, v)
    shipped = {n: v for n, v in step.items() if n.split("ELi")[-1].startswith("0E") and "ILi0E" not in n}
    assert len(shipped) == 8
    for n, v in shipped.items():
        if "k_agent_step" in n:  # the bench / training kernel against its diagnostic twin
            d = step[n.replace("ELi0E", "ELi1E")]
            assert v["private_segment_fixed_size"] <= d["private_segment_fixed_size"], (n, v, d)
            assert v["code_bytes"] < d["code_bytes"], (n, v, d)


@pytest.mark.skipif(not os.path.exists(os.path.join(codeobj.LLVM, "clang-offload-bundler")) or shutil.which("objcopy") is None,
                    reason="ROCm LLVM tools not available")
def test_march_core_keeps_its_wait_states():
    """gs_core (bdim_v2_body.h): the DPP multiply that reads the previous step's result from lane-1 must sit behind the
    two leading multiplies of the block -- they are the wait states between the VALU write of `prev` and its DPP read,
    which the hazard recogniser cannot see inside inline assembly."""
    text = codeobj.kernel_text(codeobj.disassembly(build.build()), "k_agent_stepILi2ELi0E")
    dpp = [i for i, t in enumerate(text) if t.startswith("v_mul_f32_dpp") and "row_shr:1" in t]
    assert len(dpp) >= 16  # the unrolled march loops
    for i in dpp:
        assert text[i - 1].startswith("v_mul_f32") and "dpp" not in text[i - 1], text[i - 3:i + 1]
        assert text[i - 2].startswith("v_mul_f32") and "dpp" not in text[i - 2], text[i - 3:i + 1]
