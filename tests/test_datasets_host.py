"""Input pipeline semantics, CPU: what a batch IS must equal the reference's
ImageDataset / FeatureMapDataset (dataset_loader/image_dataset.py:25-49,
feature_map_dataset.py:30-64): cv2's BGR channel order, (x - 127.5) / 127.5 in float64 then
.float(), CHW for ImageDataset and HWC for the image FeatureMapDataset returns beside a latent;
plus the additive prefetcher and packed-shard format, which must not change a single value."""
import os

import numpy as np
import pytest
import torch

from dataset_loader._tinydb_json import read_all, write_all
from dataset_loader.feature_map_dataset import FeatureMapDataset
from dataset_loader.image_dataset import ImageDataset
from dataset_loader.prefetch import DevicePrefetcher, PackedFeatureMapDataset, pack_feature_maps


def _png(path, rgb):
    from PIL import Image
    Image.fromarray(rgb).save(path)


def test_image_dataset_is_bgr_chw_and_normalised_like_cv2(tmp_path):
    rgb = np.zeros((2, 3, 3), dtype=np.uint8)
    rgb[0, 0] = (255, 0, 10)          # R, G, B
    rgb[1, 2] = (1, 128, 254)
    p = str(tmp_path / "a.png")
    _png(p, rgb)
    write_all(str(tmp_path / "d.json"), [{"image_fpath": p}])
    ds = ImageDataset(str(tmp_path / "d.json"))
    x = ds[0]
    assert x.dtype == torch.float32 and x.shape == (3, 2, 3)
    # channel 0 is BLUE (cv2.imread order), channel 2 is RED
    want = lambda v: np.float32((float(v) - 127.5) / 127.5)
    assert x[0, 0, 0] == want(10) and x[1, 0, 0] == want(0) and x[2, 0, 0] == want(255)
    assert x[0, 1, 2] == want(254) and x[1, 1, 2] == want(128) and x[2, 1, 2] == want(1)
    assert x[:, 0, 1].tolist() == [want(0)] * 3 and want(0) == -1.0 and want(255) == 1.0
    t, path = ImageDataset(str(tmp_path / "d.json"), return_filepaths=True)[0]
    assert path == p and torch.equal(t, x)
    write_all(str(tmp_path / "e.json"), [])
    with pytest.raises(Exception, match="No data found"):
        ImageDataset(str(tmp_path / "e.json"))


def test_feature_map_dataset_items_and_tinydb_layout(tmp_path):
    rng = np.random.default_rng(0)
    recs = []
    for i in range(5):
        f = str(tmp_path / f"{i}.npy")
        np.save(f, rng.standard_normal((4, 8, 8)).astype(np.float32))
        img = str(tmp_path / f"{i}.png")
        _png(img, rng.integers(0, 255, (6, 5, 3), dtype=np.uint8))
        recs.append({"fmap_path": f, "image_path": img})
    idx = str(tmp_path / "all_dataset.json")
    write_all(idx, recs)
    import json
    raw = json.load(open(idx))
    assert list(raw) == ["_default"] and list(raw["_default"]) == ["1", "2", "3", "4", "5"]  # TinyDB layout
    assert read_all(idx) == recs
    ds = FeatureMapDataset(idx)
    assert len(ds) == 5 and torch.equal(ds[3], torch.from_numpy(np.load(recs[3]["fmap_path"])))
    fm, fp, im, ip = FeatureMapDataset(idx, load_image=True, return_filepaths=True)[2]
    assert fp == recs[2]["fmap_path"] and ip == recs[2]["image_path"]
    assert im.shape == (6, 5, 3) and im.dtype == torch.float32          # HWC, as the reference returns it
    from PIL import Image
    rgb = np.asarray(Image.open(ip).convert("RGB")).astype(float)
    assert torch.equal(im, torch.from_numpy((rgb[:, :, ::-1] - 127.5) / 127.5).float())
    # packed shard: one mmap, identical items
    shape = pack_feature_maps(idx, str(tmp_path / "packed.npy"))
    assert tuple(shape) == (5, 4, 8, 8)
    pk = PackedFeatureMapDataset(str(tmp_path / "packed.npy"))
    assert len(pk) == 5 and all(torch.equal(pk[i], ds[i]) for i in range(5))
    a = [b for b in torch.utils.data.DataLoader(pk, batch_size=2, num_workers=2)]
    b = [b for b in torch.utils.data.DataLoader(ds, batch_size=2)]
    assert len(a) == 3 and all(torch.equal(x, y) for x, y in zip(a, b))


def test_prefetcher_yields_the_same_batches_in_order_on_cpu():
    data = [torch.full((2, 3), float(i)) for i in range(7)]
    got = list(DevicePrefetcher(data, "cpu", depth=3))
    assert len(got) == 7 and all(torch.equal(a, b) for a, b in zip(got, data))
    pairs = [(torch.full((2,), float(i)), torch.full((1,), float(-i))) for i in range(4)]
    got = list(DevicePrefetcher(pairs, "cpu"))
    assert all(torch.equal(g[0], p[0]) and torch.equal(g[1], p[1]) for g, p in zip(got, pairs))


@pytest.mark.gpu
def test_prefetcher_overlapped_copies_arrive_intact_on_gpu():
    g = torch.Generator().manual_seed(0)
    data = [torch.randn((64, 4, 32, 32), generator=g) for _ in range(9)]
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.cat(data)), batch_size=64,
                                         pin_memory=True)
    got = []
    for (b,) in DevicePrefetcher(loader, "cuda", depth=2):
        assert b.is_cuda
        got.append((b * 2.0).cpu())          # consumer work on the compute stream
    assert len(got) == 9 and all(torch.equal(a, 2.0 * b) for a, b in zip(got, data))
