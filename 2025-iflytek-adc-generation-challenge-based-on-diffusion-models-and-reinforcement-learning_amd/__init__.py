"""MI355X-native DDPM sampling path of the ADC molecule generator (import name: ``adcdiff``).

The directory name is not a Python identifier; ``adcdiff.py`` at the repository root loads this
package under the name ``adcdiff``.
"""
from . import _lib, ops, parallel, torch_ops
from ._lib import AdcdiffError
from .generation import MoleculeGenerator
from .models import DiffusionModel, ResidualBlock, TimeEmbedding, UNetModel



def install(package="src"):
    """Make the reference's own import statements resolve to this implementation.

    The reference's callers import the module path directly -- ``from ..models.diffusion_model import DiffusionModel``
    (src/generation/molecule_generator.py:21), ``from src.models.diffusion_model import DiffusionModel``
    (generate_submission.py:28, main.py, demo.py) -- so replacing the names in ``src/models/__init__.py`` alone would not
    reach them.  ``adcdiff.install()`` registers this package's ``models.diffusion_model`` module as
    ``<package>.models.diffusion_model`` in ``sys.modules`` (and rebinds the names on an already imported
    ``<package>.models``), so every later import of that path, absolute or relative, gets the MI355X classes.  Call it once,
    before the reference's modules are imported (e.g. first line of main.py); returns the module that was installed."""
    import sys
    from .models import diffusion_model
    sys.modules[package + ".models.diffusion_model"] = diffusion_model
    parent = sys.modules.get(package + ".models")
    if parent is not None:
        parent.diffusion_model = diffusion_model
        parent.DiffusionModel, parent.UNetModel = diffusion_model.DiffusionModel, diffusion_model.UNetModel
    return diffusion_model


__all__ = ["install", "torch_ops", "DiffusionModel", "UNetModel", "TimeEmbedding", "ResidualBlock", "AdcdiffError", "MoleculeGenerator", "ops", "parallel", "_lib"]
