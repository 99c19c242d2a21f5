"""Checkpoint I/O in the reference's ``.pth`` format (utils.py:147-203).

A checkpoint is ``torch.save({'model': state_dict, 'iteration', 'optimizer',
'learning_rate'})``.  Loading is *tolerant* exactly like the reference: every key of
the model's own state_dict takes the saved tensor if the file has it and keeps the
model's value otherwise; extra keys in the file are ignored.  Files are read with
``weights_only=True`` (tensors and plain containers only -- nothing in the file is
executed).
"""
from __future__ import annotations

import glob
import logging
import os

import torch

logger = logging.getLogger("quickvc_amd")


def load_checkpoint(checkpoint_path: str, model, optimizer=None):
    """utils.py:148-180.  Returns (model, optimizer, learning_rate, iteration)."""
    if not os.path.isfile(checkpoint_path):
        raise FileNotFoundError(checkpoint_path)
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    iteration = ckpt.get("iteration", 0)
    learning_rate = ckpt.get("learning_rate", 0.0)
    if optimizer is not None and ckpt.get("optimizer") is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    saved = ckpt["model"]
    target = model.module if hasattr(model, "module") else model
    merged = {}
    for key, own in target.state_dict().items():
        if key in saved:
            merged[key] = saved[key]
        else:
            logger.info("%s is not in the checkpoint", key)
            merged[key] = own
    target.load_state_dict(merged)
    logger.info("Loaded checkpoint '%s' (iteration %s)", checkpoint_path, iteration)
    return model, optimizer, learning_rate, iteration


def save_checkpoint(model, optimizer, learning_rate: float, iteration: int, checkpoint_path: str) -> None:
    """utils.py:183-193."""
    target = model.module if hasattr(model, "module") else model
    torch.save({"model": target.state_dict(), "iteration": iteration,
                "optimizer": None if optimizer is None else optimizer.state_dict(),
                "learning_rate": learning_rate}, checkpoint_path)


def latest_checkpoint_path(dir_path: str, regex: str = "G_*.pth") -> str:
    """utils.py:196-202: the file whose digits form the largest number."""
    files = glob.glob(os.path.join(dir_path, regex))
    files.sort(key=lambda f: int("".join(filter(str.isdigit, f))))
    return files[-1]
