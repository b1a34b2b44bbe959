"""Input pipeline for real-data throughput (additive -- the reference reads one
`.npy` per item on the training thread and copies each batch synchronously,
train_quantized_transformer.py:344-353, 407):

 - `DevicePrefetcher`: wraps any batch iterable (a DataLoader with pin_memory=True) and keeps
   `depth` batches in flight: the host->device copy of batch t+1 runs on a side HIP stream from
   pinned memory while the training step of batch t runs on the compute stream; the consumer
   waits on an event, never on the copy engine.  Same batches, same order.
 - `pack_feature_maps` / `PackedFeatureMapDataset`: every latent of a `FeatureMapDataset` index
   packed into ONE (N,C,H,W) float32 `.npy`, memory-mapped by the workers -- one sequential file
   instead of N small ones; items are bit-identical to the per-file dataset's.
"""
import collections

import numpy as np
import torch
from torch.utils.data import Dataset

from ._tinydb_json import read_all


class DevicePrefetcher:
    def __init__(self, loader, device, depth=2):
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, depth)
        self.on_gpu = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None

    def __len__(self):
        return len(self.loader)

    def _to_device(self, batch):
        if torch.is_tensor(batch):
            if not self.on_gpu:
                return batch, None
            if not batch.is_pinned():
                batch = batch.pin_memory()
            with torch.cuda.stream(self.stream):
                out = batch.to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            return out, (ev, batch)          # keep the pinned source alive until consumed
        if isinstance(batch, (list, tuple)):
            parts = [self._to_device(b) for b in batch]
            return type(batch)(p[0] for p in parts), [p[1] for p in parts]
        return batch, None

    def _wait(self, token):
        if token is None:
            return
        if isinstance(token, list):
            for t in token:
                self._wait(t)
            return
        torch.cuda.current_stream(self.device).wait_event(token[0])

    def __iter__(self):
        q = collections.deque()
        it = iter(self.loader)
        for batch in it:
            q.append(self._to_device(batch))
            if len(q) > self.depth:
                out, token = q.popleft()
                self._wait(token)
                if torch.is_tensor(out) and self.on_gpu:
                    out.record_stream(torch.cuda.current_stream(self.device))
                yield out
        while q:
            out, token = q.popleft()
            self._wait(token)
            if torch.is_tensor(out) and self.on_gpu:
                out.record_stream(torch.cuda.current_stream(self.device))
            yield out


def pack_feature_maps(dataset_path, out_path):
    """Packs every `fmap_path` of a FeatureMapDataset index (reference generate_fmap_dataset.py:
    60-72 layout) into one float32 (N,C,H,W) .npy.  Returns the shape."""
    recs = read_all(str(dataset_path))
    if len(recs) == 0:
        raise Exception("No data found.")
    first = np.load(recs[0]["fmap_path"])
    out = np.lib.format.open_memmap(str(out_path), mode="w+", dtype=np.float32,
                                    shape=(len(recs),) + first.shape)
    for i, r in enumerate(recs):
        a = np.load(r["fmap_path"])
        if a.shape != first.shape:
            raise ValueError(f"{r['fmap_path']}: shape {a.shape} != {first.shape}")
        out[i] = a
    out.flush()
    return out.shape


class PackedFeatureMapDataset(Dataset):
    """Items equal FeatureMapDataset(dataset_path)[i] (float (C,H,W)); backed by one mmap."""

    def __init__(self, packed_path):
        self.path = str(packed_path)
        self._arr = None
        self._n = np.load(self.path, mmap_mode="r").shape[0]
        if self._n == 0:
            raise Exception("No data found.")

    def __len__(self):
        return self._n

    def __getitem__(self, index):
        if self._arr is None:                 # opened lazily: once per worker process
            self._arr = np.load(self.path, mmap_mode="r")
        return torch.from_numpy(np.array(self._arr[index], dtype=np.float32))
