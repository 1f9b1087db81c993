"""Template clustering: canonical box shapes by k-medoids over 1 - IoU."""
from .cluster import centralize_bbox, compute_distances, compute_kmedoids, k_medoids  # noqa: F401
