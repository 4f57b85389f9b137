"""CPU suite: the committed fixtures are what ``make_goldens.py`` produces
from HEAD - i.e. from the reference's own ``preconditioners.py`` /
``field_split_backend.py`` run under stubs on today's producer.  Needs the
reference tree, so it runs in the build container only (the GPU box has no
``/root/reference``; nothing under ``-m gpu`` reads it)."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("FENAPACK_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "fenapack")),
                    reason="reference tree not present on this box")
def test_goldens_regenerate_from_head(tmp_path):
    out = subprocess.run([sys.executable,
                          os.path.join(HERE, "golden", "make_goldens.py"),
                          str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    committed = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
    fresh = sorted(glob.glob(str(tmp_path / "*.npz")))
    assert [os.path.basename(p) for p in committed] == \
        [os.path.basename(p) for p in fresh]
    for pc, pf in zip(committed, fresh):
        a, b = np.load(pc), np.load(pf)
        assert sorted(a.files) == sorted(b.files), pc
        for k in a.files:
            if a[k].dtype.kind in "iuSU":
                assert np.array_equal(a[k], b[k]), (pc, k)
            else:
                # inputs: round-off of the Picard state (sparse direct solve);
                # exact-solve outputs amplify it by the condition number
                tol = 1e-9 if k.endswith("_direct") else 1e-12
                scale = max(np.abs(a[k]).max(), 1e-300)
                assert a[k].shape == b[k].shape, (pc, k)
                assert np.abs(a[k] - b[k]).max() <= tol * scale, (pc, k)
        # an enclosed flow has a singular R_p: no exact-solve PCDR golden
        if "cavity" in pc:
            assert "y_RBRM1_direct" not in a.files
