"""Checkpoint discovery / loading helpers (mirror of reference utils/checkpoint.py:17-243; host-side file handling,
no device work).  File naming convention of the reference's trainer: `checkpoint_<best|final>_epoch_<N>[_<suffix>].pt`
holding the dict of `VideoToVideoDiffusion.save_checkpoint` (models/model.py:362-387)."""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import torch

logger = logging.getLogger(__name__)


def _glob_sorted(directory: str, pattern: str) -> Optional[List[Path]]:
    root = Path(directory)
    if not root.exists():
        logger.error(f"Checkpoint directory does not exist: {root}")
        return None
    return sorted(root.glob(pattern))


def _pattern(kind: str, model_suffix: Optional[str]) -> str:
    return f"checkpoint_{kind}_epoch_*_{model_suffix}.pt" if model_suffix else f"checkpoint_{kind}_epoch_*.pt"


def find_best_checkpoint(checkpoint_dir: str, model_suffix: Optional[str] = None) -> Optional[str]:
    """Last (lexicographically) `checkpoint_best_epoch_*` file, or None."""
    found = _glob_sorted(checkpoint_dir, _pattern("best", model_suffix))
    if not found:
        if found is not None:
            logger.warning(f"No checkpoints found matching pattern: {_pattern('best', model_suffix)}")
        return None
    logger.info(f"Found best checkpoint: {found[-1].name}")
    return str(found[-1])


def find_latest_checkpoint(checkpoint_dir: str, model_suffix: Optional[str] = None) -> Optional[str]:
    """A `checkpoint_final_epoch_*` file if there is one, else the best checkpoint."""
    found = _glob_sorted(checkpoint_dir, _pattern("final", model_suffix))
    if found is None:
        return None
    if found:
        logger.info(f"Found final checkpoint: {found[-1].name}")
        return str(found[-1])
    return find_best_checkpoint(checkpoint_dir, model_suffix)


def load_checkpoint(checkpoint_path: str, device: str = 'cpu', weights_only: bool = False) -> Dict:
    path = Path(checkpoint_path)
    if not path.exists():
        raise FileNotFoundError(f"Checkpoint not found: {path}")
    logger.info(f"Loading checkpoint from {path}")
    try:
        ckpt = torch.load(path, map_location=device, weights_only=weights_only)
    except Exception as exc:
        logger.error(f"Failed to load checkpoint: {exc}")
        raise
    if isinstance(ckpt, dict):
        logger.info(f"Checkpoint info: Epoch {ckpt.get('epoch', 'unknown')}, Best loss: {ckpt.get('best_loss', 'unknown')}")
    return ckpt


def load_model_from_checkpoint(model, checkpoint_path: str, device: str = 'cuda', strict: bool = True) -> Tuple[object, Dict]:
    """Load weights (new dict format or a bare state dict), move the model to `device`, switch to eval mode.
    The engine re-packs its bf16 kernel-layout weights by itself on the next call (parameter versions change)."""
    ckpt = load_checkpoint(checkpoint_path, device='cpu')
    if isinstance(ckpt, dict) and 'model_state_dict' in ckpt:
        state = ckpt['model_state_dict']
        meta = {'epoch': ckpt.get('epoch', 0), 'best_loss': ckpt.get('best_loss', float('inf')),
                'config': ckpt.get('config', {})}
    else:
        state, meta = ckpt, {}
    try:
        missing, unexpected = model.load_state_dict(state, strict=strict)
    except Exception as exc:
        logger.error(f"Failed to load model weights: {exc}")
        raise
    if missing:
        logger.warning(f"Missing keys in checkpoint: {missing}")
    if unexpected:
        logger.warning(f"Unexpected keys in checkpoint: {unexpected}")
    logger.info("Model weights loaded successfully")
    model = model.to(device)
    model.eval()
    return model, meta


def extract_model_suffix_from_path(checkpoint_path: str) -> Optional[str]:
    """'checkpoint_best_epoch_24_slice_interp_full3.pt' -> 'slice_interp_full3' (None without a suffix)."""
    parts = Path(checkpoint_path).stem.split('_')
    if 'epoch' not in parts:
        return None
    tail = parts[parts.index('epoch') + 2:]
    return '_'.join(tail) if tail else None


def list_all_checkpoints(checkpoint_dir: str, model_suffix: Optional[str] = None) -> list:
    """All `checkpoint_*` files of a directory, newest modification time first."""
    found = _glob_sorted(checkpoint_dir, f"checkpoint_*_{model_suffix}.pt" if model_suffix else "checkpoint_*.pt")
    if found is None:
        return []
    found.sort(key=lambda p: p.stat().st_mtime, reverse=True)
    logger.info(f"Found {len(found)} checkpoints")
    return [str(p) for p in found]
