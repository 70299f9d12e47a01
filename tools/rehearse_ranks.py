"""Run by hand on a GPU box (python -m pytest tools/rehearse_ranks.py -v -m gpu); kept out of tests/ because it starts process
trees and depends on the box's loopback networking, which the round-end suite should not.
The multi-rank paths of bench.py on real kernels: `python bench.py --gpus 2` starts its own two ranks, which here share the
one GPU and talk through gloo (`--backend gloo`: a rehearsal of the rank logic -- sharding by global row, the collective, the
barrier / max-over-ranks timing, the roofline window -- not a measurement; the measured runs use RCCL, one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--no-cpu-baseline"] + args, env=env,
                       capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_sample_their_shards_and_gather():
    d = run(["--batch", "256", "--timesteps", "5", "--steps", "1", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["outputs_finite"] is True and d["scaling"] == "weak"
    assert d["config"]["per_gpu_batch"] == 256 and d["config"]["global_batch"] == 512
    assert d["collective"]["bytes_result_per_rank"] == 512 * 512 * 4 and d["roofline"]["frac"] > 0


def test_two_ranks_train_data_parallel():
    """Every rank must run the roofline window's steps too: a training step holds the gradient all-reduce.
    Flaky under gloo with both ranks on one GPU: runs were seen to stop with BOTH ranks inside the same stand-alone all_reduce of
    the 177 MB device tensor (stacks from BENCH_WATCHDOG=55), after the all-reduces inside the steps had completed; with
    BENCH_PROGRESS=1 (a little stderr output per phase) the same command finishes.  Not seen in the row-sharded sampling path."""
    d = run(["--workload", "train_dp", "--batch", "256", "--steps", "2", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["outputs_finite"] is True and d["collective"]["bytes_result_per_rank"] > 170e6
