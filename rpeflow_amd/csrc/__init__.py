"""Drop-in for the reference's ``models/csrc`` package (models/csrc/__init__.py:1):
the same four names, bound to hand-written gfx950 kernels."""
from .wrapper import correlation2d, furthest_point_sampling, squared_distance, k_nearest_neighbor

__all__ = ["correlation2d", "furthest_point_sampling", "squared_distance", "k_nearest_neighbor"]
