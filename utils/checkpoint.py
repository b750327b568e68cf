"""`from utils.checkpoint import load_model_from_checkpoint, ...` keeps working (see
video-to-video-diffusion_amd/checkpoint.py)."""
import importlib

_m = importlib.import_module("video-to-video-diffusion_amd.checkpoint")
find_best_checkpoint = _m.find_best_checkpoint
find_latest_checkpoint = _m.find_latest_checkpoint
load_checkpoint = _m.load_checkpoint
load_model_from_checkpoint = _m.load_model_from_checkpoint
extract_model_suffix_from_path = _m.extract_model_suffix_from_path
list_all_checkpoints = _m.list_all_checkpoints
