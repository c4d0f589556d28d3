"""carel_vae_amd -- MI355X (gfx950) implementation of the CAREL-VAE training hot path.

The Python host code mirrors the module surface of the reference's drl_classifier_ec_mmd_final_mul.py
(`ECPEDataset`, `DrlClassifier`, `MMDStatistic`, `pdist`, `read_ECPE_data`, `train`, `save_ckp`, `load_ckp`,
`generate_self_train_data`); every numeric operation of the step is a HIP kernel in libcarel_hip.so behind the
C ABI of include/carel_hip.h.  There is no CPU / eager fallback.
`carel_vae_amd.drl_classifier_en` is the same for the reference's drl_classifier_en.py (config 4).
"""
from . import _lib  # noqa: F401
from .data import BatchLoader, PrefetchLoader, ECPEDataset, get_bow_en, get_bow_zh, read_ECPE_data  # noqa: F401
from .drl_classifier import (HSIC, DrlClassifier, FusedAdam, MMDStatistic, encoder_config, make_opt, pdist,  # noqa: F401
                             permutation_test_mat)
from . import drl_classifier_en  # noqa: F401   the three-space adversarial model of drl_classifier_en.py (same class name: DrlClassifier)
from .training import generate_self_train_data, load_ckp, save_ckp, train  # noqa: F401

__all__ = ["ECPEDataset", "BatchLoader", "PrefetchLoader", "DrlClassifier", "MMDStatistic", "pdist", "HSIC", "permutation_test_mat", "read_ECPE_data", "train",
           "generate_self_train_data", "save_ckp", "load_ckp", "get_bow_zh", "get_bow_en", "FusedAdam", "make_opt",
           "encoder_config"]
