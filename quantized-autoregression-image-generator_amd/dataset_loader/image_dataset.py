"""ImageDataset with the reference's semantics (dataset_loader/image_dataset.py:11-49):
TinyDB JSON index {"image_fpath"} -> BGR image -> (x-127.5)/127.5 -> float (C,H,W)."""
import torch
from torch.utils.data import Dataset

from ._tinydb_json import read_all
from .feature_map_dataset import _imread_bgr


class ImageDataset(Dataset):
    def __init__(self, dataset_path, return_filepaths=False):
        self.return_filepaths = return_filepaths
        self.data_list = read_all(str(dataset_path))
        if len(self.data_list) == 0:
            raise Exception("No data found.")

    def __len__(self):
        return len(self.data_list)

    def __getitem__(self, index):
        path = self.data_list[index]["image_fpath"]
        image = (_imread_bgr(path).astype(float) - 127.5) / 127.5
        tensor = torch.from_numpy(image).float().permute(2, 0, 1)
        if self.return_filepaths:
            return tensor, path
        return tensor
