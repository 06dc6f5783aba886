"""The bench line the driver parses: the committed line of the current round has every field of the contract, its numbers are
consistent with each other, and bench.py's argument parser keeps the contract's flags (CPU only, nothing is launched)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r6_bench.json")) as f:
        return json.loads(f.read().strip().split("\n")[-1])


def test_committed_bench_line_has_the_contract_fields():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1


def test_committed_bench_line_is_self_consistent():
    d = _line()
    r = d["roofline"]
    envs, sub = d["config"]["envs_per_gpu"], d["config"]["sim_steps_per_agent_step"]
    # value = env-sim-steps per second of the whole job
    assert abs(d["value"] - envs * sub * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    # achieved = 176 B per interior cell and sim-step over the launch duration measured with HIP events
    alg = 176.0 * 256 * 128 * sub * envs
    assert abs(r["achieved"] - alg / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert 1.0 < r["traffic"] / alg < 1.5  # measured HBM bytes per launch against the algorithmic figure
    # the launch average and the wall-clock step agree (one launch per step, nothing else in the timed region)
    assert abs(r["avg_launch_ms"] - d["ms_per_step"]) / d["ms_per_step"] < 0.02


def test_bench_cli_keeps_the_contract_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def test_committed_traffic_figure_belongs_to_the_shipped_kernel_sources():
    """roofline.traffic comes from profiles/<round>_hbm_traffic*.json, which record the hash of the kernel sources they
    were measured on: bench.py attaches a figure only when that hash equals the tree's (a stale figure is refused, not
    quoted).  Whenever the committed headline line and the committed traffic file belong to the same sources, the line
    must carry exactly that figure."""
    sys.path.insert(0, ROOT)
    import bench
    for key, name in bench.TRAFFIC_FILES.items():
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        tj = json.load(open(path))
        assert len(tj["kernel_src_sha16"]) == 16 and tj["hbm_bytes_per_launch"] > 0
    d = _line()
    tname = bench.TRAFFIC_FILES[(16, 100, 1024)]
    tpath = os.path.join(ROOT, "profiles", tname)
    tj = json.load(open(tpath))
    # the committed profiles belong to the committed kernel sources: whoever edits a kernel file re-runs
    # tools/refresh_profiles.sh before committing, or this fails
    assert tj["kernel_src_sha16"] == bench.kernel_source_hash(), "profiles/%s was measured on other kernel sources" % tname
    assert d["config"]["kernel_src_sha16"] == tj["kernel_src_sha16"]
    assert d["roofline"]["traffic"] == tj["hbm_bytes_per_launch"] and d["config"]["dbg"] == 0
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'tj.get("kernel_src_sha16") == kernel_source_hash()' in src  # the guard itself


def test_other_baseline_configs_are_named_not_custom():
    """the secondary workloads are lines of driver grade too: named, with cpu_baseline and (where the committed PMC
    figure belongs to the same kernel sources) roofline.traffic, >= 64 timed agent steps"""
    for f, name in (("r6_bench_cfg4.json", "BASELINE.json configs[3]"), ("r6_bench_native.json", "reference-native 384x192")):
        with open(os.path.join(ROOT, "profiles", f)) as fh:
            d = json.loads(fh.read().strip().split("\n")[-1])
        assert d["config"]["workload"].startswith(name) and d["roofline"]["frac"] > 0.35 and d["diverged_envs"] == 0
        assert d["steps"] >= 64 and "cpu_baseline" in d and d["cpu_baseline"]["kind"] == "port"
        assert d["roofline"]["traffic"] is None or 1.0 < d["roofline"]["traffic"] / d["roofline"]["algorithmic_bytes_per_launch"] < 1.5
    with open(os.path.join(ROOT, "profiles", "r6_bench_perfmode.json")) as fh:
        d = json.loads(fh.read().strip().split("\n")[-1])
    assert "PERF MODE" in d["config"]["workload"] and d["config"]["gs_order"].startswith("red-black")
