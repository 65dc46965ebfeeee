"""Random large shapes (2-12 M entries; banded, block-diagonal, power-law, hub rows, fixed-length columns) through the
library's OWN dispatch -- no layout forced -- against the CPU oracle: tools/big_shape_hunt.py, eight examples of a fixed
seed (the 224-example runs: profiles/r06_big_shape_hunt.txt)."""
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.own_row_order]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("order", ["relaxed", "strict"])
def test_large_random_shapes_match_the_oracle(gpu_required, order):
    env = dict(os.environ, PDHG_ROW_ORDER=order)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "big_shape_hunt.py"), "8", "3"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "8 examples, 0 failures" in r.stdout
