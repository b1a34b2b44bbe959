"""Image-grid writer with the behaviour of the reference's utils/image_utils.py:8-44
(BGR->RGB flip, make_grid(nrow, normalize, value_range=(-1,1)), JPEG under
<dest>/images/), without torchvision (absent here): the grid and the uint8 conversion
are restated from torchvision.utils.make_grid / save_image semantics; PIL encodes."""
import math
import os

import torch


def make_grid(images, nrow=5, padding=2, value_range=(-1, 1), pad_value=0.0):
    """(N,3,H,W) -> (3, rows*(H+pad)+pad, cols*(W+pad)+pad), values scaled to [0,1]."""
    x = images.detach().float().cpu().clone()
    lo, hi = value_range
    x = x.clamp(min=lo, max=hi).sub(lo).div(max(hi - lo, 1e-5))
    n, c, h, w = x.shape
    if n == 1:
        return x[0]
    cols = min(nrow, n)
    rows = int(math.ceil(n / cols))
    H, W = h + padding, w + padding
    grid = x.new_full((c, H * rows + padding, W * cols + padding), pad_value)
    k = 0
    for r in range(rows):
        for q in range(cols):
            if k >= n:
                break
            grid[:, r * H + padding:r * H + padding + h, q * W + padding:q * W + padding + w] = x[k]
            k += 1
    return grid


def save_images(images, file_name, dest_path, nrow=5, logging=print):
    try:
        from PIL import Image
        images = images[:, [2, 1, 0]]  # BGR (cv2 convention of the datasets) -> RGB
        grid = make_grid(images, nrow=nrow)
        folder = os.path.join(dest_path, "images")
        os.makedirs(folder, exist_ok=True)
        path = os.path.join(folder, str(file_name) + ".jpg")
        arr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
        Image.fromarray(arr).save(path)
        logging(f"Saving image: {path}")
        return True
    except Exception as e:
        logging(f"An error occured while saving image: {e}")
        return False
