"""pytest configuration: the ``gpu`` marker and import paths.

``-m "not gpu"`` runs everywhere (oracle vs golden fixtures, host logic, C-ABI symbol checks);
``-m gpu`` needs an MI355X and calls the HIP path through the C-ABI.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sfm-python_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
