"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference hot path (see oracle/ref_ops.py).

Nothing under video-to-video-diffusion_amd/, models/ or inference/ may import this package.
Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
"""
