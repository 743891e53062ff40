"""Container-only pin of the fixture chain: ``oracle/gen_golden.py`` -- which imports the REAL reference
from ``/root/reference`` and feeds it the frozen inputs defined in that script -- must regenerate every committed
``tests/golden/*.npz`` bit for bit.  Skipped where the reference tree does not exist (the GPU box)."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE = os.path.join(os.path.dirname(HERE), 'oracle')


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='reference tree only exists in the build container')
def test_gen_golden_reproduces_committed_fixtures(tmp_path):
    sys.path.insert(0, ORACLE)
    try:
        import gen_golden
        gen_golden.generate(str(tmp_path))
        diffs = gen_golden.compare_with_committed(str(tmp_path))
    finally:
        sys.path.remove(ORACLE)
        for name in [m for m in sys.modules if m == 'sunerf' or m.startswith('sunerf.')]:
            del sys.modules[name]     # the reference's package must not shadow the drop-in one for later tests
    assert not diffs, diffs
