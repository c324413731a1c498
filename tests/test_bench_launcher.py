"""Host logic of the multi-GPU bench: how `python bench.py --gpus N` starts its own ranks, which sizes and which
phase a run gets by default, and that every workload the command line accepts is one the slab driver runs."""
import argparse
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sph_amd import slab_bench  # noqa: E402


def test_default_sizes_make_eight_gpus_the_64m_dam_break():
    a = argparse.Namespace(side=None, preroll=None, workload="dam_break")
    side, total, pre = slab_bench.resolve_sizes(a, 8)
    assert (side, total) == (200, 400) and total ** 3 == 64_000_000
    assert pre == 120  # the collapse of a block four times as high takes twice as long
    assert slab_bench.resolve_sizes(a, 2)[:2] == (200, 252)
    assert slab_bench.peak_preroll(100) == 60 and slab_bench.peak_preroll(200) == 85
    a = argparse.Namespace(side=50, preroll=7, workload="sloshing")
    assert slab_bench.resolve_sizes(a, 8) == (50, 100, 7)
    a = argparse.Namespace(side=None, preroll=None, workload="uniform_cube")
    assert slab_bench.resolve_sizes(a, 4)[2] == 20


def test_launch_command_and_shared_gpu_rehearsal():
    cmd, env = slab_bench.launch_command("/x/bench.py", ["--gpus", "8", "--steps", "20"], 8, 8, 29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "8", "--steps", "20"] and env == {}
    _, env = slab_bench.launch_command("/x/bench.py", ["--gpus", "2"], 2, 1, 1)
    assert env == {"SPH_BENCH_SHARED_GPU": "1"}  # fewer GPUs than ranks: the ranks share GPU 0


def test_self_launch_starts_the_ranks_and_relays_rank_zero(tmp_path, capsys):
    """The parent's side with a stand-in for the ranks' script: two children under torch.distributed.run (gloo
    rendezvous on 127.0.0.1), rank 0 prints the JSON line, the parent relays exactly that and returns their status."""
    script = tmp_path / "ranks.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        import torch.distributed as dist
        dist.init_process_group("gloo")
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and dist.get_world_size() == 2
        print("chatter that is not the line")
        if dist.get_rank() == 0:
            print(json.dumps({"n_gpus": dist.get_world_size(), "argv": sys.argv[1:]}))
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(3 if "--fail" in sys.argv else 0)
    """))
    rc = slab_bench.self_launch(str(script), ["--gpus", "2", "--steps", "4"], 2)
    out = capsys.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1
    assert json.loads(out[0]) == {"n_gpus": 2, "argv": ["--gpus", "2", "--steps", "4"]}
    assert slab_bench.self_launch(str(script), ["--fail"], 2) != 0


def test_bench_accepts_only_workloads_the_slab_driver_runs():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "waterfall"],
                         capture_output=True, text=True)
    assert out.returncode != 0 and "invalid choice" in out.stderr
    src = open(os.path.join(ROOT, "sph_amd", "slab_bench.py")).read()
    for w in ("dam_break", "sloshing", "uniform_cube"):
        assert w in src
    assert "args.workload" in src


def test_a_hung_rank_reports_its_stacks_and_exits():
    code = "from sph_amd import slab_bench; import time; slab_bench.arm_deadline(); time.sleep(20)"
    env = dict(os.environ, SPH_BENCH_DEADLINE_S="0.5", PYTHONPATH=ROOT)
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=15)
    assert res.returncode == 1
    assert "Timeout" in res.stderr and "time.sleep" not in res.stdout
    # disarmed: the process runs to its end
    code = "from sph_amd import slab_bench; import time; slab_bench.arm_deadline(); slab_bench.disarm_deadline(); time.sleep(1)"
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=15)
    assert res.returncode == 0
