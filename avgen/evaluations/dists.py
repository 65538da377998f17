"""Reference import path avgen/evaluations/dists.py."""
from asva_amd.evaluations import frechet_distance  # noqa: F401
