"""Row sharding and the latent all-gather, exercised with two gloo ranks on the CPU.

The sampler itself has no CPU path, so the ranks 'sample' with the oracle's counter-RNG noise keyed by
global row -- the same contract DiffusionModel.p_sample_loop(row_offset=...) follows on the GPU: the
gathered result must equal the unsharded draw, for even and ragged splits.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import adcdiff
from adcdiff import parallel
from oracle import philox


def test_shard_rows_partition():
    for total in (0, 1, 7, 8, 1000, 1048576):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_rows(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        parallel.shard_rows(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, totals, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for total in totals:
            fn = lambda count, first: torch.from_numpy(philox.normal(77, philox.STREAM_XT, 0, first, count, 64)) if count else torch.empty((0, 64))
            full = parallel.sample_sharded(fn, total)
            np.save(os.path.join(out_dir, "r%d_t%d.npy" % (rank, total)), full.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sampling_equals_unsharded(world, tmp_path):
    totals = [16, 17, 1, 96]
    port = _free_port()
    mp.spawn(_worker, args=(world, port, totals, str(tmp_path)), nprocs=world, join=True)
    for total in totals:
        ref = philox.normal(77, philox.STREAM_XT, 0, 0, total, 64)
        for rank in range(world):
            got = np.load(str(tmp_path / ("r%d_t%d.npy" % (rank, total))))
            assert got.shape == ref.shape and np.array_equal(got, ref), (total, rank)


def test_generator_driver_contract():
    """The generator expands a 1-D or 1-row condition to the batch and forwards the reference's sample()
    arguments (molecule_generator.py:238-262); checked with a recording stand-in for the model."""
    calls = []

    class Recorder:
        output_dim = 8
        device = torch.device("cpu")

        def sample(self, batch_size, condition=None, use_ddim=True, ddim_steps=50, **kw):
            calls.append((batch_size, None if condition is None else tuple(condition.shape), use_ddim, ddim_steps, kw))
            return torch.zeros(batch_size, 8)

    gen = adcdiff.MoleculeGenerator({'generation_batch_size': 32})
    gen.device = torch.device("cpu")
    gen.set_diffusion_model(Recorder())
    out = gen.generate_initial_features(70, torch.ones(8), seed=5)
    assert out.shape == (70, 8)
    assert [c[0] for c in calls] == [32, 32, 6]
    assert calls[0][1] == (32, 8) and calls[2][1] == (6, 8)
    assert all(c[2] is True and c[3] == 50 for c in calls)
    assert [c[4]["row_offset"] for c in calls] == [0, 32, 64]
    calls.clear()
    gen._generate_batch_features(4, torch.ones(1, 8))
    assert calls[0][1] == (4, 8)
    gen._generate_batch_features(4, None)
    assert calls[1][1] is None
    with pytest.raises(ValueError):
        adcdiff.MoleculeGenerator({}).generate_initial_features(1)


def _grad_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(n)) for n in (5, 1000, 37, 4096)] + [torch.nn.Parameter(torch.zeros(3))]
        for i, p in enumerate(params[:-1]):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        parallel.allreduce_gradients(params, bucket_bytes=4200)      # several buckets; the last parameter has no gradient
        np.save(os.path.join(out_dir, "g%d.npy" % rank), np.concatenate([p.grad.numpy().ravel() for p in params[:-1]]))
        assert params[-1].grad is None
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_averages_over_ranks(tmp_path):
    world = 2
    mp.spawn(_grad_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    expect = np.concatenate([np.full(n, 1.5 * (i + 1), dtype=np.float32) for i, n in enumerate((5, 1000, 37, 4096))])
    for rank in range(world):
        assert np.array_equal(np.load(str(tmp_path / ("g%d.npy" % rank))), expect)
