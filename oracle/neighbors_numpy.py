"""TEST INFRASTRUCTURE ONLY -- numpy specification of the neighbour list, restating the reference's
own executable spec tests/test_neighbors.py:14-71 (lexsort ordering :14-16, triclinic minimum image
z -> y -> x :19-27, O(N^2) pair enumeration per molecule + cutoff filter :30-71)."""
import numpy as np


def sort_neighbors(neighbors, deltas, distances):
    order = np.lexsort(neighbors)
    return neighbors[:, order], deltas[order], distances[order]


def apply_pbc(deltas, box):
    if box is None:
        return deltas
    b = np.asarray(box, dtype=deltas.dtype)
    deltas = deltas - np.outer(np.round(deltas[:, 2] / b[2, 2]), b[2])
    deltas = deltas - np.outer(np.round(deltas[:, 1] / b[1, 1]), b[1])
    deltas = deltas - np.outer(np.round(deltas[:, 0] / b[0, 0]), b[0])
    return deltas


def reference_neighbors(pos, batch, loop, include_transpose, cutoff, box, cutoff_lower=0.0):
    """pos [N,3], batch [N] (sorted) -> (neighbors [2,P], deltas [P,3], distances [P]) lexsorted."""
    pos = np.asarray(pos)
    batch = np.asarray(batch)
    counts = np.bincount(batch)
    offs = np.concatenate([[0], np.cumsum(counts)])
    blocks = [np.stack(np.tril_indices(int(c), -1)) + offs[m] for m, c in enumerate(counts)]
    nb = np.concatenate(blocks, axis=1) if blocks else np.zeros((2, 0), dtype=np.int64)
    if box is not None and np.asarray(box).ndim == 3:
        d = np.concatenate([apply_pbc(pos[b[0]] - pos[b[1]], np.asarray(box)[m]) for m, b in enumerate(blocks)])
    else:
        d = apply_pbc(pos[nb[0]] - pos[nb[1]], box)
    dist = np.linalg.norm(d, axis=-1)
    keep = (dist < cutoff) & (dist >= cutoff_lower)
    nb, d, dist = nb[:, keep], d[keep], dist[keep]
    if include_transpose:
        nb = np.concatenate([nb, nb[::-1]], axis=1)
        d = np.concatenate([d, -d])
        dist = np.concatenate([dist, dist])
    if loop:
        ar = np.arange(pos.shape[0])
        nb = np.concatenate([nb, np.stack([ar, ar])], axis=1)
        d = np.concatenate([d, np.zeros((pos.shape[0], 3), dtype=d.dtype)])
        dist = np.concatenate([dist, np.zeros(pos.shape[0], dtype=dist.dtype)])
    return sort_neighbors(nb, d, dist)
