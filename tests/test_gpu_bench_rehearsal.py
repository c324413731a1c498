"""The multi-GPU leg of bench.py as far as one GPU allows: started plainly with --gpus 2 it launches its own two ranks
(which share the GPU and say so), and with one rank the library's own transport (RCCL issued from C++) drives the same
loop.  Both must print one well-formed JSON line and lose no particle."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(*argv):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                         timeout=280, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("workload", ["dam_break", "sloshing"])
def test_two_ranks_started_plainly_on_one_gpu(workload):
    out = run_bench("--gpus", "2", "--side", "40", "--steps", "6", "--warmup", "2", "--preroll", "30",
                    "--workload", workload, "--no-cpu-baseline")
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 6
    assert out["config"]["workload"] == f"{workload}_50^3" and out["config"]["particles"] == 125000
    assert sum(out["config"]["particles_per_rank"]) == 125000
    assert "REHEARSAL" in out["config"]["parallelism"]
    assert out["value"] > 0 and out["roofline"]["kernel"] == "k_force" and 0 < out["roofline"]["frac"] < 1
    assert out["phase"]["timed_steps"] == [32, 38]


def test_one_rank_through_the_library_transport():
    out = run_bench("--gpus", "1", "--force-slab", "--side", "50", "--steps", "6", "--warmup", "2", "--preroll", "30")
    assert out["n_gpus"] == 1 and out["config"]["particles"] == 125000
    assert "RCCL send/recv issued by the library" in out["config"]["parallelism"]
    assert out["exchange"]["remainder_rounds_per_rank"] == [0]
    assert out["cpu_baseline"]["kind"] in ("reference", "port") and out["cpu_baseline"]["cores"] >= 1
    assert out["rest_density_extra"]["value"] > 0
