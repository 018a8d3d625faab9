import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def product_lib():
    """libmatchtigs.so, built in-tree (build() compiles it; it must exist for every test that touches the product)."""
    from matchtigs_amd import _lib

    if not _lib.LIB_PATH.exists():
        import __graft_entry__ as ge

        ge.build()
    return _lib.load()
