import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Every Srs / default Basis a test creates holds a digit-multiple table sized by this budget (GB per SRS half; the library's
# default is 48 and a quarter of the free memory -- 56 GB per k = 13 SRS -- and bench.py's service profile 160: 189 GB).  The suite keeps several SRS alive at once, so it runs on a smaller budget -- the
# same kernels with narrower digits (k = 13: 9 bits instead of 15); test_gpu_parity.py::test_msm_table_path forces the
# wide ones.
os.environ.setdefault("ZKFHE_TABLE_GB", "4")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
