import ctypes
import os
import subprocess
import sys

import pytest

try:  # torch bundles its own HIP runtime: it must be loaded before libppg_hip.so pulls in /opt/rocm's, or torch.cuda sees no GPU
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "practical-path-guiding_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

ORACLE_SO = os.path.join(ROOT, "oracle", "libppg_oracle.so")
HIP_SO = os.path.join(PKG, "lib", "libppg_hip.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    if not (os.path.exists(ORACLE_SO) and os.path.exists(HIP_SO)):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "__graft_entry__.py")])


@pytest.fixture(scope="session")
def oracle_lib():
    _ensure_built()
    return ctypes.CDLL(ORACLE_SO)


@pytest.fixture(scope="session")
def hip_lib_path():
    _ensure_built()
    return HIP_SO


CBOX_PROPS = dict(budgetType="spp", maxDepth=10, rrDepth=10, strictNormals=1)  # scenes/cbox/cbox.xml:9-11
IMPROVED = dict(sampleCombination="inversevar", bsdfSamplingFractionLoss="kl", spatialFilter="stochastic",
                directionalFilter="box", sTreeThreshold=4000, sppPerPass=1)  # README.md:30-37 / cbox-improved.xml


def make_oracle(lib, threads=8, acc=0, adam=0, **props):
    import ppg_host
    e = ppg_host.Engine(lib, "ppgo_", **props)
    # (the oracle's passes are OpenMP loops over 32x32 blocks: on the GPU box's 256 hardware threads a test film of a few blocks made
    # 250 threads wait for six — the region-round tests took 49 s each, 0.5 s with 16 threads; PPG_TEST_ORACLE_THREADS overrides the cap)
    threads = max(1, min(int(threads), int(os.environ.get("PPG_TEST_ORACLE_THREADS", "32"))))
    lib.ppgo_set_modes(e.ctx, acc, adam, threads)
    return e
