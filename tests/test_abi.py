"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, exports every
symbol include/afc_env.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from tests.conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from rlfluidcontrol_amd import build, env
    build.build()
    return env.load_library()


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "afc_env.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(afc_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(lib):
    from rlfluidcontrol_amd import env
    decl = _declared_symbols()
    assert len(decl) >= 25
    assert sorted(env.ABI_SYMBOLS) == decl  # the python binding tracks the header
    for name in decl:
        assert hasattr(lib, name), name


def test_default_config_is_reference_native(lib):
    from rlfluidcontrol_amd import env
    cfg = env.default_config()
    # clientCFD.pde:5-14,94-96
    assert (cfg.resolution, cfg.x_lengths, cfg.y_lengths, cfg.reynolds) == (24, 16, 8, 500)
    assert cfg.substeps == 16 and cfg.init_time == 1.0 and cfg.end_time == 50.0 and cfg.act_scale == 5.0
    assert abs(cfg.t_step - 0.0075) < 1e-9 and abs(cfg.dR - 0.125) < 1e-9 and abs(cfg.gR - 0.2) < 1e-7
    assert abs(cfg.reward_penalty - 0.18675613444715936) < 1e-7  # server.py:64
    assert C.sizeof(env.AfcConfig) == 4 * 24


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rlfluidcontrol_amd import env
    with pytest.raises(env.AfcError) as ei:
        env.AFCEnv(num_envs=1, resolution=8, reynolds=100)
    assert "no HIP device" in str(ei.value) or "-2" in str(ei.value)


def test_bad_arguments_rejected(lib):
    from rlfluidcontrol_amd import env
    h = C.c_void_p()
    assert lib.afc_create(None, C.byref(h)) == -1
    cfg = env.default_config(num_envs=0)
    assert lib.afc_create(C.byref(cfg), C.byref(h)) == -1
    assert b"num_envs" in lib.afc_last_error()
