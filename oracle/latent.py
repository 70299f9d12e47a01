"""CPU restatement of what the reference's latent -> SMILES heuristic computes from a latent row.  Test infrastructure only.

Follows src/generation/molecule_generator.py of the reference, read as text (the module cannot be imported here: it imports
rdkit at module level and rdkit is not installed), so the string logic below is **parity unpinned** by execution; the
arithmetic (np.sum / np.mean of a float32 row) IS the reference's own, because it is numpy's.
"""
import numpy as np

# molecule_generator.py:1140-1156, in order
DRUG_LIKE_SMILES = ["CCO", "CC(=O)O", "c1ccccc1", "c1ccc(O)cc1", "c1ccc(N)cc1", "CCN", "CC(C)O", "CCCC", "c1ccc(C)cc1", "CC(=O)N",
                    "c1ccc(Cl)cc1", "CCc1ccccc1", "CC(C)(C)O", "c1ccc2ccccc2c1", "CC(=O)c1ccccc1"]


def summary(features):
    """(mean, sum, f[3], f[5], f[8]) of one float32 row, as :1080-1081, :1114-1126 and :1160 read them."""
    f = np.asarray(features, dtype=np.float32)
    return np.array([np.mean(f), np.sum(f), f[3], f[5], f[8]], dtype=np.float32)


def summaries(latents):
    return np.stack([summary(r) for r in np.asarray(latents, dtype=np.float32)])


def pairwise_sum_f32(a):
    """numpy's float32 pairwise summation written out (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum, unit
    stride): what np.sum does on a contiguous float32 vector; used to document the order the GPU kernel reproduces."""
    a = np.asarray(a, dtype=np.float32)
    n = a.shape[0]
    if n < 8:
        res = np.float32(0.0)
        for v in a:
            res = np.float32(res + v)
        return res
    if n <= 128:
        r = [np.float32(a[j]) for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = np.float32(r[j] + a[i + j])
            i += 8
        res = np.float32(np.float32(np.float32(r[0] + r[1]) + np.float32(r[2] + r[3])) + np.float32(np.float32(r[4] + r[5]) + np.float32(r[6] + r[7])))
        while i < n:
            res = np.float32(res + a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return np.float32(pairwise_sum_f32(a[:n2]) + pairwise_sum_f32(a[n2:]))


def decode_candidates(s, n_features=512):
    """(primary, fallback) SMILES strings from one summary row: _decode_features_to_smiles + _add_functional_groups
    (:1072-1130) and _generate_fallback_smiles (:1132-1163).  The reference returns `primary` when RDKit validates it and
    `fallback` otherwise; validation stays with the caller."""
    mean, total, f3, f5, f8 = (float(v) for v in s)
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
    idx = int(abs(total) * len(DRUG_LIKE_SMILES)) % len(DRUG_LIKE_SMILES)
    return primary, DRUG_LIKE_SMILES[idx]
