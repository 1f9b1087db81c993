"""Offline template clustering with the reference's call surface (tinyfaces/clustering/cluster.py:13-130,
tinyfaces/clustering/k_medoids.py:6-72): ground-truth boxes are reduced to their shape (centred at the origin), pairwise
distance = 1 - IoU, k-medoids picks the canonical shapes (`templates.json` ships the result for WIDER FACE, so this only runs for
a new dataset: tinyfaces/datasets/__init__.py:24-38).

Only the reference's own `local` k-medoids is provided natively (alternate nearest-medoid assignment and per-cluster medoid
update until the medoid set is stable, same np.random.choice seeding); the `pyclustering` / `pyclust` options of the reference
delegate to third-party libraries that are not part of this stack.  The distance matrix is one vectorised numpy expression
instead of the reference's n^2 Python loop (cluster.py:29-37)."""
import warnings

import numpy as np


def centralize_bbox(bboxes):
    """(x1, y1, x2, y2) -> (-(w-1)/2, -(h-1)/2, (w-1)/2, (h-1)/2) with w = x2 - x1 + 1 (cluster.py:13-25): position is dropped."""
    b = np.asarray(bboxes, dtype=np.float64)
    half_w = (b[:, 2] - b[:, 0] + 1 - 1) / 2
    half_h = (b[:, 3] - b[:, 1] + 1 - 1) / 2
    return np.stack([-half_w, -half_h, half_w, half_h], axis=1)


def compute_distances(bboxes, device=None):
    """1 - jaccard_index for every pair (cluster.py:28-37 with tinyfaces/metrics.py:8-40: plain areas, no +1, IoU 0 when the
    union is not positive).  device='cuda': the n^2 matrix comes from the HIP kernel (tf_pairwise_iou_distance, bit-exact with
    this expression), which is what the 5000-box default of compute_kmedoids (25 M distances) wants; None: numpy on the host."""
    if device is not None and str(device) != "cpu":
        from .. import ops
        return ops.pairwise_iou_distance(bboxes, device=device).cpu().numpy()
    b = np.asarray(bboxes, dtype=np.float64)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    xa = np.maximum(b[:, None, 0], b[None, :, 0]); ya = np.maximum(b[:, None, 1], b[None, :, 1])
    xb = np.minimum(b[:, None, 2], b[None, :, 2]); yb = np.minimum(b[:, None, 3], b[None, :, 3])
    inter = (xb - xa) * (yb - ya)
    union = area[:, None] + area[None, :] - inter
    iou = np.divide(inter, union, out=np.zeros_like(inter), where=union > 0)
    return 1 - iou


def k_medoids(distances, k, rng=np.random):
    """k_medoids.py:6-72: random distinct start, then assign / update until no medoid moves.  Returns (medoid indices, cluster id
    of every point)."""
    n = distances.shape[0]
    medoids = rng.choice(n, size=k, replace=False)
    previous = np.zeros(k)
    while not np.all(medoids == previous):
        previous = np.copy(medoids)
        member = np.argmin(distances[medoids, :], axis=0)
        for c in range(k):
            inside = member == c
            if not inside.any():
                warnings.warn("Cluster {} is empty!".format(c))
                continue
            current = distances[medoids[c], inside].sum()
            costs = distances[np.ix_(inside, inside)].sum(axis=1)
            best = int(np.argmin(costs))
            if costs[best] < current:
                medoids[c] = np.nonzero(inside)[0][best]
    return medoids, member


def compute_kmedoids(bboxes, cls, option="local", indices=15, max_clusters=35, max_limit=5000, rng=np.random, device=None):
    """cluster.py:40-130 for option='local': one clustering per k in [indices, max_clusters].  The returned list keeps the
    reference's shape: `indices` empty dicts first, then one entry per k, so entry k sits at index k when the caller passes
    indices == max_clusters == k, which is how tinyfaces/datasets/__init__.py:26-33 reads clustering[num_templates]."""
    if option != "local":
        raise NotImplementedError(f"compute_kmedoids(option={option!r}): only the reference's own 'local' k-medoids is built in; "
                                  "'pyclustering' / 'pyclust' need those third-party packages")
    clustering = [{} for _ in range(indices)]
    shapes = centralize_bbox(bboxes)
    if shapes.shape[0] > max_limit:
        shapes = shapes[rng.choice(np.arange(shapes.shape[0]), size=max_limit, replace=False)]
    dist = compute_distances(shapes, device=device)
    for k in range(indices, max_clusters + 1):
        medoids, _ = k_medoids(dist, k, rng)
        clustering.append({"n_clusters": k, "medoids": [shapes[m, :] for m in medoids], "class": cls})
    return clustering
