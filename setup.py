"""Packaging of qsparse_amd: `pip install .` compiles the HIP library for gfx950 (hipcc, no GPU needed) and ships it
inside the package; `python -c "import __graft_entry__ as g; g.build()"` does the same in place for development."""
import os
import sys

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


class BuildWithHip(build_py):
    def run(self):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as entry

        entry.build_hip()          # qsparse_amd/libqsparse_hip.so, picked up through package_data below
        super().run()


def version():
    for line in open(os.path.join(ROOT, "qsparse_amd", "__init__.py")):
        if line.startswith("__version__"):
            return line.split('"')[1]
    raise RuntimeError("no __version__")


setup(
    name="qsparse-amd",
    version=version(),
    description="MI355X-native (gfx950) quantize/prune operators with the API of mlzxy/qsparse",
    packages=["qsparse_amd"],
    package_data={"qsparse_amd": ["libqsparse_hip.so", "csrc/*.h", "csrc/*.hip"]},
    python_requires=">=3.8",
    install_requires=["torch>=1.9.0", "numpy"],
    cmdclass={"build_py": BuildWithHip},
)
