"""Builds the PyTorch-ROCm extension module `HCSPMM` (the reference's module name,
hybrid_kernel/setup.py:4-14 there) in-tree:  python setup.py build_ext --inplace   (or `install`,
as the reference README says).  Host-only C++ (CppExtension, so torch's hipify step never runs);
the gfx950 kernels live in ../csrc/libhcspmm.so, built by ../csrc/Makefile with hipcc.
"""
import os
import subprocess

from setuptools import setup
from torch.utils.cpp_extension import BuildExtension, CppExtension

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(HERE, "..", "csrc"))
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "..", "include"))
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

if not os.path.exists(os.path.join(CSRC, "libhcspmm.so")):
    subprocess.check_call(["make", "-C", CSRC, "-j", "8"])

setup(
    name="HCSPMM",
    ext_modules=[
        CppExtension(
            "HCSPMM", ["hybrid_all.cpp"],
            include_dirs=[INCLUDE, os.path.join(ROCM, "include")],
            define_macros=[("__HIP_PLATFORM_AMD__", "1"), ("USE_ROCM", "1")],
            library_dirs=[CSRC],
            libraries=["hcspmm", "c10_hip"],
            extra_compile_args=["-O2", "-std=c++17"],
            extra_link_args=["-Wl,-rpath,$ORIGIN/../csrc", "-Wl,-rpath," + CSRC],
        )
    ],
    cmdclass={"build_ext": BuildExtension},
)
