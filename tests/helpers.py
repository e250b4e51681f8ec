"""Shared helpers for the parity tests."""
import ctypes as C
import hashlib

import numpy as np


def to_oracle_params(p):
    """SphParams (product) -> OracleParams (checker); the two structs have identical layout
    (asserted in test_capi.py)."""
    from oracle.oracle import OracleParams
    o = OracleParams()
    assert C.sizeof(o) == C.sizeof(p)
    C.memmove(C.byref(o), C.byref(p), C.sizeof(o))
    return o


def to_product_params(o):
    from smoothed_particle_hydrodynamics_amd import SphParams
    p = SphParams()
    assert C.sizeof(o) == C.sizeof(p)
    C.memmove(C.byref(p), C.byref(o), C.sizeof(p))
    return p


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def live_mask(counts, cap):
    """boolean mask over the n*cap list storage selecting the stored entries"""
    return (np.arange(cap)[None, :] < counts[:, None]).ravel()


def max_rel(a, b, floor=0.0):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = np.maximum(np.maximum(np.abs(a), np.abs(b)), floor)
    den[den == 0] = 1.0
    return float((np.abs(a - b) / den).max()) if a.size else 0.0


def vec_rel(a, b):
    """per-particle relative error of 3-vectors: |a-b| / max(|a|,|b|)"""
    a = np.asarray(a, np.float64).reshape(-1, 3)
    b = np.asarray(b, np.float64).reshape(-1, 3)
    num = np.linalg.norm(a - b, axis=1)
    den = np.maximum(np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1))
    den[den == 0] = 1.0
    return num / den
