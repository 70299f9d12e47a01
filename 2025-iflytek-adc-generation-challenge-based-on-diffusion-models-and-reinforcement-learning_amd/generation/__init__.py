from .molecule_generator import MoleculeGenerator

__all__ = ["MoleculeGenerator"]
