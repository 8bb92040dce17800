#!/usr/bin/env python
"""Build tools/libgemm_tune.so (variants of the product GEMM template + the round-1 snapshot) in the build container; it travels
to the GPU box with the snapshot.  `python tools/gemm_tune.py --build`"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "libgemm_tune.so")


def build():
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           os.path.join(ROOT, "tools", "gemm_tune.hip"), "-o", SO])


def main():
    """Build the tuner library; timing lives in tools/gemm_ab.py (order-unbiased A/B with clock warm-up and phase stamps)."""
    build()
    print("built %s -- time variants with: python tools/gemm_ab.py 38,37,25 [--gn 0,1] [--k 512,1024] [--stamps]" % SO)


if __name__ == "__main__":
    main()
