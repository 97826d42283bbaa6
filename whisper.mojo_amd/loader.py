"""Mirror of /root/reference/loader.mojo: WeightLoader(filename) reads the headerless fp32 file."""
from __future__ import annotations

import numpy as np


class WeightLoader:
    """loader.mojo:5-31.  The file image is handed whole to the device library (which validates its size —
    the reference does not, loader.mojo:21-27); next_tensor stays available for host-side inspection."""

    def __init__(self, filename: str):
        self.filename = filename
        self.raw_data = np.fromfile(filename, dtype=np.float32)  # raises like the reference (loader.mojo:10-11)
        self.size = self.raw_data.size
        self.offset = 0

    @classmethod
    def from_array(cls, weights: np.ndarray) -> "WeightLoader":
        self = cls.__new__(cls)
        self.filename = None
        self.raw_data = np.ascontiguousarray(weights, np.float32).ravel()
        self.size = self.raw_data.size
        self.offset = 0
        return self

    def next_tensor(self, rows: int, cols: int) -> np.ndarray:
        count = rows * cols
        if self.offset + count > self.size:
            raise ValueError("weight file exhausted")
        t = self.raw_data[self.offset:self.offset + count].reshape(rows, cols).copy()
        self.offset += count
        return t
