"""Checkpoint I/O with the reference's on-disk contract (reference utils/model_utils.py:
6-52): a plain dict written with torch.save to <dest>/models_checkpoint/<file_name>;
load_model returns (ok, dict-on-CPU).  Dict schemas are the callers' (SURVEY.md 5)."""
import os

import torch


def save_model(dest_path, file_name, model_dict, logging=print):
    try:
        folder = os.path.join(dest_path, "models_checkpoint")
        os.makedirs(folder, exist_ok=True)
        torch.save(model_dict, os.path.join(folder, file_name))
        return True
    except Exception as e:  # the reference swallows and reports
        logging(f"Exception occured while saving model: {e}.")
        return False


def load_model(checkpoint_path, logging=print, weights_only=True):
    """Loads with torch's weights-only unpickler by default: every checkpoint dict of this
    pipeline (tensors, numbers, strings, tuples, None, an optimizer state_dict) -- and of the
    reference, whose schemas are the same -- loads under it, and nothing in a downloaded
    `--model-path` file can execute code.  weights_only=False is an explicit opt-out for a
    legacy file the safe loader refuses."""
    if not os.path.exists(checkpoint_path):
        logging("Checkpoint does not exist.")
        return False, None
    ckpt = torch.load(checkpoint_path, map_location=torch.device("cpu"), weights_only=weights_only)
    return True, ckpt
