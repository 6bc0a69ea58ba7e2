import os
import sys

import pytest

# Every library workspace starts as 0xFF bytes (NaN as a float, -1 as an index) while the tests run: a kernel that reads
# scratch nobody wrote fails a test instead of passing on whatever the memory pool held before.  (This is how the stale
# tail of the approximate EMD's dense candidate list was found: 0 * exp2(NaN coordinates) is NaN, not 0.)  Read once when
# the library allocates its first workspace; child processes of the tests inherit it.
os.environ.setdefault('PCC_WS_POISON', '1')
# arms include/pcc_test_hooks.h (inert otherwise): the auction's failure-reporting path is driven through it
os.environ.setdefault('PCC_TEST_HOOKS', '1')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (container only; skipped elsewhere)')


@pytest.fixture(scope='session')
def oracle_mod():
    import oracle

    oracle.build()
    oracle.set_threads(min(8, oracle.max_threads()))
    return oracle


@pytest.fixture(scope='session')
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


@pytest.fixture(scope='session', autouse=True)
def poison_fresh_outputs():
    """The host layer hands the kernels ``torch.empty`` outputs (the reference's contract: freshly allocated, fully
    overwritten).  While the tests run, Python-level ``torch.empty`` / ``empty_like`` on the GPU return NaN (-1 for integer
    types) instead of recycled memory, so an output element a kernel forgets to write cannot pass a comparison by luck."""
    import torch

    if not torch.cuda.is_available():
        yield
        return
    orig_empty, orig_like = torch.empty, torch.empty_like

    def poison(t):
        if t.is_cuda and t.numel():
            if t.is_floating_point():
                t.fill_(float('nan'))
            elif t.dtype in (torch.int32, torch.int64, torch.int16, torch.uint8):
                t.fill_(255 if t.dtype == torch.uint8 else -1)
        return t

    def empty(*args, **kwargs):
        return poison(orig_empty(*args, **kwargs))

    def empty_like(*args, **kwargs):
        return poison(orig_like(*args, **kwargs))

    torch.empty, torch.empty_like = empty, empty_like
    try:
        yield
    finally:
        torch.empty, torch.empty_like = orig_empty, orig_like
