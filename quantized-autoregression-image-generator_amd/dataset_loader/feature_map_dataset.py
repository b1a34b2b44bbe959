"""FeatureMapDataset with the reference's semantics (dataset_loader/
feature_map_dataset.py:14-64): TinyDB JSON index -> one .npy latent per item -> float
tensor (C,H,W); optionally the source image (BGR, (x-127.5)/127.5, HWC)."""
import numpy as np
import torch
from torch.utils.data import Dataset

from ._tinydb_json import read_all


def _imread_bgr(path):
    from PIL import Image  # cv2 is absent here; PIL decodes, channels flipped to cv2's BGR
    return np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1]


class FeatureMapDataset(Dataset):
    def __init__(self, dataset_path, load_image=False, return_filepaths=False):
        self.load_image = load_image
        self.return_filepaths = return_filepaths
        self.data_list = read_all(str(dataset_path))
        if len(self.data_list) == 0:
            raise Exception("No data found.")

    def __len__(self):
        return len(self.data_list)

    def __getitem__(self, index):
        rec = self.data_list[index]
        fmap_path = rec["fmap_path"]
        with open(fmap_path, "rb") as f:
            fmap = torch.from_numpy(np.load(f)).float()
        if self.load_image:
            image_path = rec["image_path"]
            image = torch.from_numpy((_imread_bgr(image_path).astype(float) - 127.5) / 127.5).float()
            if self.return_filepaths:
                return fmap, fmap_path, image, image_path
            return fmap, image
        if self.return_filepaths:
            return fmap, fmap_path
        return fmap
