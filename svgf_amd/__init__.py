"""svgf_amd — MI355X-native SVGF denoiser (hot path of jacquespillet/SVGF, src/Filter.cuh).

The product is the C-ABI library built from svgf_amd/csrc (include/svgf.h); this package is the
Python host-side mirror used by tests, bench.py and the multi-GPU strip runner.
"""
__version__ = "0.1.0"
