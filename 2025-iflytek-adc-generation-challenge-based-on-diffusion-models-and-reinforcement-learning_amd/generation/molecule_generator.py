"""The sampling driver of the reference's MoleculeGenerator (src/generation/molecule_generator.py:155-271).

Only the part that drives the diffusion model is mirrored: batching by ``generation_batch_size``, expanding
a 1-row protein condition to the batch, calling ``DiffusionModel.sample`` (DDIM, 50 steps by default) under
``no_grad``.  Everything downstream of the latents in the reference (latent -> SMILES heuristic, RDKit
validation and scoring, the RL edit loop, :273-2278) is CPU cheminformatics outside this package.
"""
import logging
import time
from typing import Dict, List, Optional

import torch

from ..models.diffusion_model import DiffusionModel


class MoleculeGenerator:
    """Holds a DiffusionModel and produces latent feature batches the way the reference's generator does."""

    def __init__(self, config: Dict):
        self.config = config
        self.logger = logging.getLogger(__name__)
        self.max_attempts = config.get('max_attempts', 1000)
        self.batch_size = config.get('generation_batch_size', 32)          # molecule_generator.py:41
        self.use_mixed_precision = config.get('use_mixed_precision', True)  # :43
        self.use_ddim = config.get('use_ddim', True)                        # the reference hard-codes True (:253, :260)
        self.ddim_steps = config.get('ddim_steps', 50)
        self.device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self.diffusion_model: Optional[DiffusionModel] = None

    def load_models(self, diffusion_path: Optional[str] = None, rl_path: Optional[str] = None):
        """molecule_generator.py:117-135 (the RL agent is outside this package)."""
        if diffusion_path:
            self.diffusion_model = DiffusionModel(self.config.get('diffusion', {}))
            self.diffusion_model.load_model(diffusion_path)
        if rl_path:
            raise NotImplementedError("the RL agent is not part of the diffusion hot path")

    def set_diffusion_model(self, diffusion_model: DiffusionModel):
        """molecule_generator.py:146-153."""
        self.diffusion_model = diffusion_model

    def _expand_condition(self, batch_size: int, protein_features: Optional[torch.Tensor]):
        """molecule_generator.py:238-245."""
        if protein_features is None:
            return None
        condition = protein_features.to(self.device)
        if condition.dim() == 1:
            condition = condition.unsqueeze(0).repeat(batch_size, 1)
        elif condition.shape[0] == 1:
            condition = condition.repeat(batch_size, 1)
        return condition

    def _generate_batch_features(self, batch_size: int, protein_features: Optional[torch.Tensor] = None, **kw) -> torch.Tensor:
        """The tensor half of _generate_batch_molecules (molecule_generator.py:232-262): [batch, output_dim] latents."""
        if self.diffusion_model is None:
            raise ValueError("diffusion model not loaded")
        with torch.no_grad():
            condition = self._expand_condition(batch_size, protein_features)
            return self.diffusion_model.sample(batch_size=batch_size, condition=condition, use_ddim=self.use_ddim,
                                               ddim_steps=self.ddim_steps, **kw)

    def generate_initial_features(self, num_molecules: int, protein_features: Optional[torch.Tensor] = None,
                                  seed: Optional[int] = None) -> torch.Tensor:
        """Latents for num_molecules samples in batches of generation_batch_size, the loop of
        generate_initial_molecules (molecule_generator.py:155-230) without the SMILES stage."""
        if self.diffusion_model is None:
            raise ValueError("diffusion model not loaded")
        start = time.time()
        chunks: List[torch.Tensor] = []
        done = 0
        while done < num_molecules:
            current = min(self.batch_size, num_molecules - done)
            kw = {} if seed is None else dict(seed=seed, row_offset=done)
            chunks.append(self._generate_batch_features(current, protein_features, **kw))
            done += current
        self.logger.info("generated %d latent vectors in %.2fs", num_molecules, time.time() - start)
        return torch.cat(chunks, dim=0)

    def _features_to_smiles(self, features):
        raise NotImplementedError("latent -> SMILES decoding needs RDKit and is outside the diffusion hot path")
