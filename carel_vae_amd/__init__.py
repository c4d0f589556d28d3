"""carel_vae_amd -- MI355X (gfx950) implementation of the CAREL-VAE training hot path.

Python host code mirrors the module surface of the reference's drl_classifier_ec_mmd_final_mul.py;
every numeric operation on the step path is a HIP kernel in libcarel_hip.so (C ABI: include/carel_hip.h).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
