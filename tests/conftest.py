import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The suite runs ~100 tests in ONE process: dozens of contexts, multi-member provers with eight contexts each, RCCL, both curves,
# transforms up to 2^28.  Every hardware queue of the process reserves scratch for the largest frame launched on it; with 16 queues
# three of five runs of this process ended in HSA_STATUS_ERROR_OUT_OF_RESOURCES (264 GB of device memory free) late in the suite,
# with 4 or 8 none ever did (profiles/r5_q16_suite_abort.txt).  8 is also what libzkhip asks for by itself.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`: the oracle, the host logic, the kernels on the fibre emulator — single-threaded work, 20 minutes in
    one process) runs in four worker processes when pytest-xdist is there and the command line did not choose (`-n ...`):
    under five minutes on eight cores, same tests, same `-x`.  The GPU suite NEVER does: its tests share one device and its memory.
    ZKHIP_TEST_WORKERS=0 (or `-n 0`) keeps one process; any other number chooses the worker count."""
    if os.environ.get("PYTEST_XDIST_WORKER") or getattr(config.option, "numprocesses", None) is not None:
        return None
    if "not gpu" not in (getattr(config.option, "markexpr", "") or "") or config.getoption("collectonly", False) or config.getoption("usepdb", False):
        return None
    try:
        workers = int(os.environ.get("ZKHIP_TEST_WORKERS", "4"))
        from xdist.plugin import pytest_cmdline_main as xdist_cmdline_main
    except Exception:
        return None
    if workers <= 1 or (os.cpu_count() or 1) < 4 or not config.pluginmanager.hasplugin("xdist"):
        return None
    config.option.numprocesses = workers
    xdist_cmdline_main(config)      # (derives --dist load / --tx popen x workers from the count, as `-n` on the command line would have)
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer CPU test")


# The hot-path parity file (fields, NTT, MSM, witness map, full proofs: SURVEY.md §8a rows K1-K9) runs first, then the
# other files in the order below, so that under `-x` a failure in a peripheral feature can never hide the hot path.
_FILE_ORDER = ["test_gpu_parity.py", "test_oracle_c.py", "test_oracle_py.py", "test_abi.py", "test_emu_kernels.py", "test_multi_device.py",
               "test_gm17.py", "test_ingest.py", "test_ingest_reference_shapes.py", "test_poseidon.py", "test_random_circuits.py", "test_formats.py"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        return _FILE_ORDER.index(name) if name in _FILE_ORDER else len(_FILE_ORDER)
    items.sort(key=rank)      # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _vram_used_gib():
    import json
    import subprocess
    try:
        txt = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--json"], capture_output=True, text=True, timeout=30).stdout
        d = next(iter(json.loads(txt[txt.index("{"):]).values()))
        return [int(v) for k, v in d.items() if "Used" in k][0] / 2 ** 30
    except Exception:
        return None


@pytest.fixture(autouse=True)
def _device_memory_log(request):
    """ZKHIP_TEST_MEMLOG=<file>: device memory in use after every test (rocm-smi), one line per test — which test leaves how much
    behind in the suite's one process (an out-of-resources abort late in the suite is the sum of its predecessors)."""
    yield
    path = os.environ.get("ZKHIP_TEST_MEMLOG")
    if path:
        u = _vram_used_gib()
        with open(path, "a") as f:
            f.write("%8.2f GiB  %s\n" % (u if u is not None else -1.0, request.node.nodeid))
