"""`python bench.py --gpus N` starts its own ranks when no rendezvous is in the environment (the command shape the round-end
driver uses).  Run here on the CPU with the launcher self-test workload: gloo, no GPU, no native library."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def test_bench_starts_its_own_ranks_and_prints_one_line():
    p, lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "launcher_selftest", "--batch", "1001"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["gathered_rows_match"] is True          # ragged shards (501 + 500 rows) gathered in global row order


def test_three_ranks_and_single_process():
    p, lines = _run(["--gpus", "3", "--steps", "1", "--warmup", "0", "--workload", "launcher_selftest", "--batch", "10"])
    assert p.returncode == 0 and json.loads(lines[0])["ranks"] == 3, p.stderr[-2000:]
    p, lines = _run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--workload", "launcher_selftest"])
    assert p.returncode == 0 and json.loads(lines[0])["ranks"] == 1, p.stderr[-2000:]


def test_mismatched_world_size_is_refused():
    p, _ = _run(["--gpus", "2", "--workload", "launcher_selftest"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "must agree" in p.stderr


def test_parent_of_a_multi_gpu_run_imports_neither_torch_nor_the_library():
    """The launching process must not initialise a GPU: it may not even import torch."""
    code = ("import sys, runpy, subprocess\n"
            "subprocess.run = lambda cmd, env=None: type('R', (), {'returncode': 0})()\n"
            "sys.argv = ['bench.py', '--gpus', '2', '--workload', 'launcher_selftest']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    assert e.code == 0, e.code\n"
            "assert 'torch' not in sys.modules and 'adcdiff' not in sys.modules, 'launcher imported torch'\n" % os.path.join(ROOT, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
