"""Builds raytracetorch_amd/_fastparams (host-only C++, no device code) in place:
    python raytracetorch_amd/csrc/setup_fastparams.py build_ext --inplace      (from the repository root)"""
import os

from setuptools import setup
from torch.utils.cpp_extension import BuildExtension, CppExtension

HERE = os.path.dirname(os.path.abspath(__file__))
setup(name="raytracetorch_amd_fastparams",
      ext_modules=[CppExtension("raytracetorch_amd._fastparams", [os.path.join(HERE, "fastparams.cpp")],
                                extra_compile_args=["-O2", "-std=c++17"])],
      cmdclass={"build_ext": BuildExtension.with_options(use_ninja=False)},
      script_args=None)
