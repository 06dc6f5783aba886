"""bench.py's N>1 path end to end on CPU: the same spawn / rendezvous / barrier / gather / max-over-ranks / reporting
code the driver's multi-GPU run executes, with gloo and a stub in place of the HIP env (`--dry-run-cpu`)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus2_dry_run_prints_one_line_with_both_ranks():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--steps", "4",
                          "--warmup", "1", "--envs", "8", "--repeats", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak"
    t = d["timing"]
    assert len(t["ms_per_step_per_rank"]) == 2 and all(x > 0 for x in t["ms_per_step_per_rank"])
    assert t["gather_ms_per_step"] > 0 and len(t["ms_per_step_all_passes"]) == 2
    # the slowest rank defines the step (the stub's rank 1 sleeps twice as long as rank 0)
    assert d["ms_per_step"] >= max(t["ms_per_step_per_rank"]) * 0.9
    # value = env sim-steps of ALL ranks per second
    assert abs(d["value"] - 2 * 8 * 16 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.02
    assert "dry run" in d["config"]["workload"] and d["roofline"]["traffic"] is None
