"""Mirror of the reference's src/models/__init__.py exports for the diffusion path."""
from .diffusion_model import DiffusionModel, ResidualBlock, TimeEmbedding, UNetModel

__all__ = ["DiffusionModel", "UNetModel", "TimeEmbedding", "ResidualBlock"]
