"""Build + install (see pyproject.toml).  The native step is devis_amd.build: hipcc --offload-arch=gfx950 on csrc/*.hip ->
devis_amd/libmsda_hip.so, which is then shipped as package data (there is no CPU build: the reference's setup.py:36-47 refuses
to build without CUDA in the same way)."""
import os
import shutil
import sys

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


class BuildWithHip(build_py):
    def run(self):
        sys.path.insert(0, ROOT)
        from devis_amd import build as hip_build
        hip_build.ensure()                                      # raises when hipcc is missing and no library is there
        inc = os.path.join(ROOT, "devis_amd", "include")        # the header travels inside the package (build.include_dir)
        os.makedirs(inc, exist_ok=True)
        shutil.copy2(os.path.join(ROOT, "include", "msda.h"), os.path.join(inc, "msda.h"))
        try:
            super().run()
        finally:
            shutil.rmtree(inc, ignore_errors=True)


setup(
    name="devis-amd",
    version="0.6.0",       # (also in pyproject.toml: setuptools < 61 does not read the [project] table)
    packages=["devis_amd", "devis_amd.functions", "devis_amd.modules"],
    package_data={"devis_amd": ["libmsda_hip.so", "libmsda_hip.srchash", "csrc/*.hip", "csrc/*.h", "csrc/*.inc", "include/*.h", "routes.json"]},
    py_modules=["MultiScaleDeformableAttention"],
    package_dir={"": "integration", "devis_amd": "devis_amd"},
    cmdclass={"build_py": BuildWithHip},
)
