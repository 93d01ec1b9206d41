"""Comparison helpers for the parity tests."""
import numpy as np


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bits_equal(a, b, what=""):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    bad = (bits(a) != bits(b)) & ~(np.isnan(a) & np.isnan(b))  # a NaN is a NaN: sign and payload are not part of the contract
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} fp32 bit patterns differ"


def assert_knn_tie_aware(idx, dist, ref_idx, ref_dist, ref_next=None, what=""):
    """sorted distance lists identical bit for bit; indices identical
    wherever the reference distance is unique within the row's top-(k+1); inside a
    group of equal distances any order / either boundary candidate is accepted."""
    idx, ref_idx = np.asarray(idx, np.int64), np.asarray(ref_idx, np.int64)
    assert idx.shape == ref_idx.shape, f"{what}: shape {idx.shape} vs {ref_idx.shape}"
    assert_bits_equal(dist, ref_dist, what + " sorted distances")
    d = np.asarray(ref_dist, np.float32)
    k = d.shape[-1]
    tied = np.zeros(d.shape, bool)
    if k > 1:
        eq = d[..., 1:] == d[..., :-1]
        tied[..., 1:] |= eq
        tied[..., :-1] |= eq
    if ref_next is not None:
        tied[..., -1] |= np.asarray(ref_next, np.float32) == d[..., -1]
    else:
        tied[..., -1] = True  # boundary unknown: do not judge the last slot
    wrong = (idx != ref_idx) & ~tied
    assert not wrong.any(), f"{what}: {int(wrong.sum())} untied indices differ (of {idx.size}; {int(tied.sum())} tied)"
    return int(tied.sum())
