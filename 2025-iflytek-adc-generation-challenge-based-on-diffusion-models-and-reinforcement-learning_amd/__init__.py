"""MI355X-native DDPM sampling path of the ADC molecule generator (import name: ``adcdiff``).

The directory name is not a Python identifier; ``adcdiff.py`` at the repository root loads this
package under the name ``adcdiff``.
"""
from . import _lib, ops, parallel
from ._lib import AdcdiffError
from .generation import MoleculeGenerator
from .models import DiffusionModel, ResidualBlock, TimeEmbedding, UNetModel

__all__ = ["DiffusionModel", "UNetModel", "TimeEmbedding", "ResidualBlock", "AdcdiffError", "MoleculeGenerator", "ops", "parallel", "_lib"]
