"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

from conftest import v3d

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + '.npz'))
    return {k: d[k] for k in d.files}


def weights_checksum(sd):
    return float(sum(v.double().abs().sum().item() * (i + 1)
                     for i, (k, v) in enumerate(sorted(sd.items()))))


def golden_costreg_weights(g):
    """Re-create the seeded weights a fixture was generated with and verify their checksum."""
    syn = v3d('synthetic')
    sd = syn.costregnet_weights(seed=int(g['weights_seed']), sharpen=float(g['sharpen']))
    cs = weights_checksum(sd)
    assert abs(cs - float(g['weights_checksum'])) <= 1e-9 * abs(cs), \
        'seeded weights drifted from the ones the golden fixture was generated with'
    return sd


def t(x):
    return torch.from_numpy(np.asarray(x))
