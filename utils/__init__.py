"""Drop-in for the reference's `utils` package (utils/__init__.py:1-8): same import paths, MI355X engine underneath."""
from .logger import setup_logger
from .metrics import calculate_psnr, calculate_ssim

__all__ = ['setup_logger', 'calculate_psnr', 'calculate_ssim']
