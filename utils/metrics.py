"""`from utils.metrics import calculate_video_metrics` keeps working: device-side PSNR / SSIM (see
video-to-video-diffusion_amd/metrics.py)."""
import importlib

_m = importlib.import_module("video-to-video-diffusion_amd.metrics")
calculate_psnr = _m.calculate_psnr
calculate_ssim = _m.calculate_ssim
calculate_video_metrics = _m.calculate_video_metrics
