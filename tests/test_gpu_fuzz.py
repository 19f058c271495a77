"""GPU: short randomised runs of the two differential checkers (the long runs are recorded in
profiles/r02_micro.txt): alternative kernels against the plain ones, remap family against the
oracle."""
import importlib.util
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(path, argv):
    spec = importlib.util.spec_from_file_location('_fuzz_' + os.path.basename(path)[:-3], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = sys.argv
    sys.argv = [path] + [str(a) for a in argv]
    try:
        return mod.main()
    finally:
        sys.argv = old


def test_alternative_kernels_have_the_bits_of_the_plain_ones():
    assert _run(os.path.join(ROOT, 'tools', 'fuzz_paths.py'), [30, 11]) == 0


def test_remap_family_against_the_oracle_random_cases(oracle):
    assert _run(os.path.join(ROOT, 'tests', 'fuzz_oracle.py'), [60, 12]) == 0
