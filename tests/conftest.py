import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "pangenome-index_amd"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def golden():
    return GOLDEN


@pytest.fixture(scope="session")
def built():
    """Both shared libraries, built in-tree if missing (hipcc cross-compiles without a GPU)."""
    import oracle_ffi
    import pgx_ffi

    if not os.path.exists(os.path.join(ROOT, "oracle", "libpgx_oracle.so")):
        oracle_ffi.build()
    if not os.path.exists(pgx_ffi.LIB_PATH):
        pgx_ffi.build()
    return True


@pytest.fixture(scope="session")
def workdir(tmp_path_factory, built):
    return str(tmp_path_factory.mktemp("pgx_work"))


@pytest.fixture(scope="session")
def x_index(workdir, golden):
    """encoded .ri + synthetic compact tags built from test_data/x.rl_bwt (BASELINE configs[0..1])."""
    import pgx_workload

    ri, tags = pgx_workload.build_index_from_rlbwt(os.path.join(golden, "x.rl_bwt"), workdir, "x")
    return ri, tags


@pytest.fixture(scope="session")
def xy_paths(golden):
    d = os.path.join(golden, "bidirectional_test")
    return os.path.join(d, "xy.ri"), os.path.join(d, "xy_bidirectional_compressed.tags")
