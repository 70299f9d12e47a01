"""The sampling driver of the reference's MoleculeGenerator (src/generation/molecule_generator.py:155-271).

Only the part that drives the diffusion model is mirrored: batching by ``generation_batch_size``, expanding
a 1-row protein condition to the batch, calling ``DiffusionModel.sample`` (DDIM, 50 steps by default) under
``no_grad``, and the reduction of a latent row to the five numbers the reference's latent -> SMILES heuristic
reads (:1072-1163), done on the GPU so that 20 bytes per sample are copied to the host instead of the row.
Everything that needs RDKit (validation, scoring, the RL edit loop, :273-2278) is CPU cheminformatics outside
this package.
"""
import logging
import time
from typing import Dict, List, Optional

import torch

from .. import ops
from ..models.diffusion_model import DiffusionModel

# molecule_generator.py:1140-1156, in order
DRUG_LIKE_SMILES = ("CCO", "CC(=O)O", "c1ccccc1", "c1ccc(O)cc1", "c1ccc(N)cc1", "CCN", "CC(C)O", "CCCC", "c1ccc(C)cc1", "CC(=O)N",
                    "c1ccc(Cl)cc1", "CCc1ccccc1", "CC(C)(C)O", "c1ccc2ccccc2c1", "CC(=O)c1ccccc1")


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
        """molecule_generator.py:238-245: one protein's features for the whole batch.  The reference repeats the row; here it is
        expanded (same shape and values, row stride 0), which lets sample() see that the batch shares one condition: the
        sampling loops then apply the time projection of t_emb + c_emb as a per-timestep row bias instead of a per-row operand."""
        if protein_features is None:
            return None
        condition = protein_features.to(self.device)
        if condition.dim() == 1:
            condition = condition.unsqueeze(0).expand(batch_size, -1)
        elif condition.shape[0] == 1:
            condition = condition.expand(batch_size, -1)
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

    def latent_summaries(self, features: torch.Tensor) -> torch.Tensor:
        """[N, D] latents on the GPU -> [N, 5] float32 (mean, sum, f[3], f[5], f[8]) per row, in numpy's float32 summation
        order: everything _decode_features_to_smiles / _add_functional_groups / _generate_fallback_smiles
        (molecule_generator.py:1072-1163) read of a row."""
        return ops.latent_summary(features.to(torch.float32))

    @staticmethod
    def smiles_candidates(summary_row, n_features: int = 512):
        """(primary, fallback) strings for one summary row.  primary = _add_functional_groups(base) over the base chosen
        by the mean thresholds (molecule_generator.py:1084-1126); fallback = the drug-like table entry selected by
        int(|sum| * 15) % 15 (:1158-1161).  The reference returns primary when RDKit validates it and fallback otherwise
        (:1062-1066); that check stays with the caller.  The reference's functional-group tests are three separate `if`s
        (:1110-1126); each taken branch returns, so the `elif` chain below selects the same string.  No execution of the
        reference pins this heuristic (it imports rdkit, which is not installed): parity unpinned."""
        mean, total, f3, f5, f8 = (float(v) for v in summary_row)
        if mean > 0.5:
            base = "c1ccccc1"
        elif mean > 0.3:
            base = "CCCCCC"
        elif mean > 0.1:
            base = "CCO"
        else:
            base = "CC"
        primary = base
        if n_features > 10:
            if f5 > 0.7:
                primary = "c1ccc(O)cc1" if "c1ccccc1" in base else base + "O"
            elif f8 > 0.6:
                primary = "c1ccc(N)cc1" if "c1ccccc1" in base else base + "N"
            elif f3 > 0.8:
                primary = base + "C(=O)O"
        return primary, DRUG_LIKE_SMILES[int(abs(total) * len(DRUG_LIKE_SMILES)) % len(DRUG_LIKE_SMILES)]

    def _features_to_smiles(self, features):
        raise NotImplementedError("validated latent -> SMILES decoding needs RDKit and is outside the diffusion hot path; "
                                  "latent_summaries + smiles_candidates give its RDKit-free part")
