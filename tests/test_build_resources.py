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
    # both kernels: CPL = 0 (reference kernels, diagnostic unit) + CPL = 1..4 x {shipped, diagnostics, perf mode} +
    # CPL = 1, 2 of afc::lat (the up-legs marching down the rows: grids of up to 128 columns)
    assert len(step) == 2 * (1 + 4 * 3 + 2), sorted(step)
    for n, v in step.items():
        assert v["vgpr_count"] + v["agpr_count"] <= 128, (n, v)
        # MI355X_MICROARCH.md: 256-thread blocks per CU <= floor(800 / (ceil(sgpr/16)*16 + 16))
        assert 800 // (-(-v["sgpr_count"] // 16) * 16 + 16) >= 4, (n, v)
        if "ILi0E" not in n:  # optimised kernels: static LDS small, the dynamic part (tiles / rings) gets the rest of 40 KB
            assert v["group_segment_fixed_size"] <= 3 * 1024, (n, v)
            # spill budget (bytes per lane): what is spilled sits outside the innermost loops (tools/loop_scan.py)
            # (the projection tail is a function of its own since round 4: its callee-saved registers are part of the figure;
            # the diagnostic instantiation, which nobody times, gets a little more room)
            # (afc::lat and afc::diag call mg_up_tr / tm_march, functions of their own: the callee-saved registers of the two
            # nested calls are part of the figure -- saved once per call, six calls per sim-step)
            diag = n.split("ELi")[-1].startswith("1E")
            lat = n.split("ELi")[-1].startswith("3E")
            assert v["private_segment_fixed_size"] <= ((1152 if (diag or lat) else 640) if "k_agent_step" in n else 1280), (n, v)
    shipped = {n: v for n, v in step.items() if n.split("ELi")[-1].startswith("0E") and "ILi0E" not in n}
    assert len(shipped) == 8
    for n, v in shipped.items():
        if "k_agent_step" in n:  # the bench / training kernel against its diagnostic twin
            d = step[n.replace("ELi0E", "ELi1E")]
            # (within the allocator's noise: the two instantiations answer any source change by a few spill slots either way;
            # the instantiations for one-tile grids also carry the fused advection advdif_v3f, which their diagnostic
            # twins do not -- its row loop is spill-free, tools/loop_scan.py)
            fused = "ILi1E" in n or "ILi2E" in n
            assert v["private_segment_fixed_size"] <= d["private_segment_fixed_size"] + (64 if fused else 32), (n, v, d)
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


@pytest.mark.skipif(not os.path.exists(os.path.join(codeobj.LLVM, "clang-offload-bundler")) or shutil.which("objcopy") is None,
                    reason="ROCm LLVM tools not available")
def test_base_priorities_are_where_they_were_measured():
    """prio_rotate / prio_by_mate (bdim_v2_body.h): the shipped kernels of the one-tile grids (<1,0>, <2,0>) and of the
    192-column grids (<3,0>) set base priorities 1 and 2 and read their base back (s_getreg_b32 of STATUS.USER_PRIO) around
    every march; the 256-column kernels read it back but never set one (rotation +6 % there); afc::lat and afc::perf
    (<*,3>, <*,2>) carry round 5's statements verbatim: s_setprio 3 / 0 only, no s_getreg."""
    dis = codeobj.disassembly(build.build())

    def prios(name):
        text = codeobj.kernel_text(dis, name)
        levels = {int(x.split()[1], 0) for x in text if x.startswith("s_setprio")}
        return levels, sum(1 for x in text if x.startswith("s_getreg_b32") and "HW_REG_STATUS" in x)

    for k in ("k_agent_stepILi1ELi0E", "k_agent_stepILi2ELi0E", "k_agent_stepILi3ELi0E"):
        levels, reads = prios(k)
        assert levels == {0, 1, 2, 3} and reads >= 2, (k, levels, reads)
    levels, reads = prios("k_agent_stepILi4ELi0E")
    assert reads >= 2, reads  # (its marches return to whatever the wave had: 0)
    for k in ("k_agent_stepILi1ELi3E", "k_agent_stepILi2ELi3E", "k_agent_stepILi1ELi2E", "k_agent_stepILi2ELi2E"):
        levels, reads = prios(k)
        assert levels <= {0, 3} and reads == 0, (k, levels, reads)
        assert levels == {0, 3} or k.endswith("Li2E"), (k, levels)  # (the red-black kernels have no serial stretch on most grids)


@pytest.mark.skipif(not os.path.exists(os.path.join(codeobj.LLVM, "clang-offload-bundler")) or shutil.which("objcopy") is None,
                    reason="ROCm LLVM tools not available")
def test_every_lds_dma_sets_m0_right_in_front_of_itself():
    """lds_dma16 (bdim_v2_body.h) sets m0 inside inline assembly; LLVM ignores clobbers of reserved registers, so nothing
    but this check keeps a value the COMPILER put into m0 (it uses m0 for its own LDS-DMA copies, mg_bottom's table load)
    from being overwritten by the block before its use -- or the other way round.  In every kernel of the built
    library: (1) every mention of m0 is an `s_mov_b32 m0, sN`; (2) every global_load_lds finds such a write at most 8
    instructions in front of it with no branch, no branch TARGET and no other LDS-DMA in between (so no path reaches
    a copy with an m0 set for another one, hoisted or stale); (3) no instruction that reads m0 implicitly in another way
    (LDS-direct / GWS / sendmsg / movrel / gpr-index / interpolation, buffer loads to LDS) exists."""
    import re
    implicit = re.compile(r"^(ds_gws|s_sendmsg|v_interp|s_movrel|v_movrel|s_set_gpr_idx|ds_\w*addtid|buffer_load\w*\s.*\blds\b|lds_direct)")
    import subprocess
    n_dma = 0
    for co in codeobj.extract(build.build()):  # (addresses restart in every code object: one translation unit at a time)
        n_dma += _check_m0_discipline(subprocess.check_output([os.path.join(codeobj.LLVM, "llvm-objdump"), "-d", co], text=True), implicit)
    assert n_dma > 500  # (every step kernel carries 30-130 of them)


def _check_m0_discipline(disasm, implicit):
    import re
    ins = []  # (address, text) of every instruction, in order
    for l in disasm.split("\n"):
        m = re.match(r"\s+(\S.*?)\s+// ([0-9A-Fa-f]+):", l)
        if m:
            ins.append((int(m.group(2), 16), m.group(1).strip()))
    targets = set()
    for a, t in ins:
        m = re.match(r"s_c?branch\S*\s+(\d+)", t)
        if m:
            off = int(m.group(1))
            targets.add(a + 4 + 4 * (off - 65536 if off >= 32768 else off))
    n_dma = 0
    for i, (a, t) in enumerate(ins):
        assert not implicit.match(t), t
        if re.search(r"\bm0\b", t):
            assert t.startswith("s_mov_b32 m0, s"), t
        if t.startswith("global_load_lds"):
            n_dma += 1
            ok = False
            for k in range(i - 1, max(i - 9, -1), -1):
                ak, tk = ins[k]
                if tk.startswith("s_mov_b32 m0, s"):
                    ok = all(ins[q][0] not in targets for q in range(k + 1, i + 1))
                    break
                assert not tk.startswith(("s_branch", "s_cbranch", "global_load_lds", "s_setpc", "s_swappc")), [x[1] for x in ins[k:i + 1]]
            assert ok, [x[1] for x in ins[max(i - 9, 0):i + 1]]
    return n_dma
