import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionfinish(session, exitstatus):
    """Write every parity error the GPU tests measured (tests.helpers.record) to a JSON file."""
    import json
    import os
    try:
        from tests.helpers import PARITY_LOG
    except Exception:
        return
    if not PARITY_LOG:
        return
    path = os.environ.get("VDMI_PARITY_OUT") or os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_errors.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump({"note": "relative L2 error ||hip - ref|| / ||ref|| per GPU parity test, against the bar the test holds it to",
                       "errors": PARITY_LOG}, f, indent=1, sort_keys=True)
    except OSError:
        pass
