"""The checkpoint format of the reference (diffusion_model.py:521-543) against a file the reference itself wrote:
tests/golden/reference_checkpoint_h16.pth (a 16-wide model after one train_step, written by the reference's save_model in
tests/golden/make_checkpoint.py).  The kernels do not run at that width; reading and writing a checkpoint does not
involve them, so this runs on the CPU with strict_shapes=False."""
import os

import numpy as np
import pytest
import torch

import adcdiff
from conftest import GOLDEN

CKPT = os.path.join(GOLDEN, "reference_checkpoint_h16.pth")
SMALL = dict(input_dim=16, hidden_dim=16, output_dim=16, strict_shapes=False, memory_monitor_enabled=False)


def test_reference_checkpoint_loads_unchanged(tmp_path):
    g = np.load(os.path.join(GOLDEN, "tiny_ckpt.npz"))
    ref = torch.load(CKPT, map_location="cpu", weights_only=True)
    m = adcdiff.DiffusionModel(dict(SMALL))
    m.load_model(CKPT)                                        # strict load: every key and shape must match
    sd = m.model.state_dict()
    assert list(sd.keys()) == list(ref["model_state_dict"].keys()) and len(sd) == int(g["h16.n_tensors"])
    for k, v in ref["model_state_dict"].items():
        assert torch.equal(sd[k].cpu(), v), k
    assert abs(sum(float(v.double().sum()) for v in sd.values()) - float(g["h16.param_checksum"])) < 1e-9
    a, b = m.optimizer.state_dict(), ref["optimizer_state_dict"]
    assert a["param_groups"] == b["param_groups"]
    assert len(a["state"]) == int(g["h16.opt_n_state"]) and float(a["state"][0]["step"]) == float(g["h16.opt_step"]) == 1.0
    for i, st in b["state"].items():
        for k in ("step", "exp_avg", "exp_avg_sq"):
            assert torch.equal(a["state"][i][k].cpu(), st[k]), (i, k)
    # what save_model writes has the reference's structure, key order and values
    out = str(tmp_path / "ours.pth")
    m.save_model(out)
    mine = torch.load(out, map_location="cpu", weights_only=True)
    assert list(mine.keys()) == list(ref.keys())
    assert list(mine["model_state_dict"].keys()) == list(ref["model_state_dict"].keys())
    assert mine["optimizer_state_dict"]["param_groups"] == b["param_groups"]
    assert all(torch.equal(mine["model_state_dict"][k], ref["model_state_dict"][k]) for k in ref["model_state_dict"])
    assert bool(g["ours_loaded_by_reference"])               # and the reference restored such a file (recorded by make_checkpoint.py)


def test_unsupported_widths_fail_at_construction():
    """The kernels take widths that are multiples of 64 with input_dim == output_dim; anything else must fail when the model is
    built, with a message that says so, not at the first forward."""
    for bad in (dict(input_dim=100, hidden_dim=128, output_dim=100), dict(input_dim=128, hidden_dim=96, output_dim=128),
                dict(input_dim=128, hidden_dim=256, output_dim=64)):
        with pytest.raises(ValueError, match="multiples of 64|input_dim == output_dim"):
            adcdiff.DiffusionModel(dict(bad, memory_monitor_enabled=False))
        with pytest.raises(ValueError):
            adcdiff.UNetModel(bad["input_dim"], bad["hidden_dim"], bad["output_dim"], 1000)
    adcdiff.DiffusionModel(dict(input_dim=128, hidden_dim=128, output_dim=128, memory_monitor_enabled=False))
    adcdiff.DiffusionModel(dict(input_dim=100, hidden_dim=96, output_dim=100, strict_shapes=False, memory_monitor_enabled=False))
