"""Screen-space sharding for multi-GPU rendering: interleaved row bands (band b -> rank b % N) so
that cheap sky bands and expensive mesh bands spread evenly; every rank writes its bands compactly
and rank 0 reassembles the frame after ONE gather of equal-sized shards (padding the short ones).
The scene is replicated per GPU; there is no other data-path collective."""
import numpy as np

BAND_ROWS = 8


def shard_rows(height, band_rows, shard, n_shards):
    n_bands = (height + band_rows - 1) // band_rows
    return sum(min(band_rows, height - b * band_rows) for b in range(shard, n_bands, n_shards))


def max_shard_rows(height, band_rows, n_shards):
    return max(shard_rows(height, band_rows, s, n_shards) for s in range(n_shards))


def shard_row_map(height, band_rows, shard, n_shards):
    """global row index of every compact local row of `shard`."""
    rows = []
    n_bands = (height + band_rows - 1) // band_rows
    for b in range(shard, n_bands, n_shards):
        y0 = b * band_rows
        rows.extend(range(y0, min(y0 + band_rows, height)))
    return np.asarray(rows, dtype=np.int64)


def assemble(shards, height, width, band_rows):
    """shards: list (one per rank) of arrays (>= shard_rows, width, 4).  Returns (height, width, 4).
    Works on numpy arrays or torch tensors (index assignment only)."""
    n = len(shards)
    first = shards[0]
    if hasattr(first, "new_zeros"):
        import torch
        out = first.new_zeros((height, width, 4))
        for s in range(n):
            idx = torch.as_tensor(shard_row_map(height, band_rows, s, n), device=first.device)
            out[idx] = shards[s][: len(idx)]
        return out
    out = np.zeros((height, width, 4), dtype=first.dtype)
    for s in range(n):
        idx = shard_row_map(height, band_rows, s, n)
        out[idx] = shards[s][: len(idx)]
    return out
