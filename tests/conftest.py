import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the product library is a build artefact (git-ignored): compile it when a fresh checkout
    # runs the tests before __graft_entry__.build() (hipcc cross-compiles gfx950 without a GPU)
    import subprocess
    lib = os.path.join(ROOT, "stereoreconstruction_amd", "libstereo_recon_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "stereoreconstruction_amd", "csrc"), "-j4"],
                              stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def hip_ctx():
    """One srh_context on device 0 for the whole GPU session (fails loudly without a GPU)."""
    # some GPU tests hand torch tensors to the library: torch's bundled ROCm runtime has to attach
    # to the GPU before the library's (see capi._torch_runtime_first), so import it here
    import torch  # noqa: F401
    from stereoreconstruction_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
