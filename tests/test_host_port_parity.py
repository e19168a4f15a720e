"""CPU tier: the product's Python plumbing + the kernels' exact arithmetic (host build of
bbd_math.h) against the reference's golden vectors.  No GPU needed."""
import pytest
import torch

from golden_io import Case, DIRECT_CASES
from fused_runner import run_direct_case, compare_with_golden, compare_grads
from host_port import HostPortBackend


@pytest.fixture(scope="module")
def backend():
    return HostPortBackend()


def test_constant_divisions_are_correctly_rounded(backend):
    # every 97th float bit pattern across the positive range the kernels can see
    assert backend.check_div(0x2f000000, 8_000_000, 97) == 0
    assert backend.check_div(0xaf000000, 2_000_000, 389) == 0


@pytest.mark.parametrize("name", DIRECT_CASES)
def test_fused_path_host_port(name, backend):
    case = Case(name)
    tr, inputs, outputs, losses = run_direct_case(case, backend)
    report = compare_with_golden(case, tr, outputs, losses, exact=True)
    losses["loss"].backward()
    compare_grads(case, report)
