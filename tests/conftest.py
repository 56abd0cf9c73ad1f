import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))  # tests/routes.py

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The built library is git-ignored: a fresh checkout has only sources.  Build it (hipcc cross-compiles gfx950
    # without a GPU) so that the suite does not depend on __graft_entry__.build() having run first.
    if not os.path.exists(os.path.join(ROOT, "panoptikon_amd", "libpvs.so")):
        import subprocess

        subprocess.check_call([sys.executable, "-m", "panoptikon_amd.build"], cwd=ROOT, stdout=subprocess.DEVNULL)


def pytest_collection_modifyitems(config, items):
    # A test that never comes back (a collective with a missing rank, a device that stopped answering) must fail, not hold the
    # whole run: every test gets a generous ceiling unless it set its own (pytest-timeout; absent plugin: no ceiling).
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
