"""No exception leaves an ``extern "C"`` function (SURVEY 8(b) "Errors": every
export returns a status, "never throws across the ABI, never aborts"; the
reference turns the exceptions of its PCPYTHON context into PETSc error codes:
fenapack/field_split.py:135-140).

* every export of the three sources is a function-try-block closed by the
  catch macro (a static check of the sources);
* a ``pcdh_*`` export whose allocation fails (a child process under
  ``RLIMIT_AS``) returns ``PCDH_ERR_NOMEM`` - the child lives, no signal;
* ``libpcd_host`` built with ASan + UBSan runs ``tests/test_host_native.py``;
* (GPU) ``pcd_set_csr`` under the same address-space limit returns a status
  code with a message and the handle can still be destroyed.
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fenapack_amd", "csrc")


def _exports(path, prefix):
    """(name, has function-try-block) of every non-trivial ``int`` export."""
    text = open(path).read()
    out = []
    for m in re.finditer(r"^int (%s\w+)\(([^;{]*?)\)\s*(try\s*)?\{([^\n]*)$"
                         % prefix, text, flags=re.M | re.S):
        one_liner = m.group(4).strip().endswith("}")
        out.append((m.group(1), bool(m.group(3)), one_liner))
    return out


def test_every_export_is_a_function_try_block():
    seen = 0
    for f, prefix, skip in (("pcd_abi.hip", "pcd_", ()),
                            ("pcd_setup.hip", "pcd_", ()),
                            ("pcd_krylov.hip", "pcd_", ()),
                            ("pcd_producer_abi.hpp", "pcd_", ()),
                            ("pcd_apply.hip", "pcd_", ("pcd_apply_dev",)),
                            ("pcd_host.cpp", "pcdh_", ())):
        text = open(os.path.join(CSRC, f)).read()
        for name, guarded, one_liner in _exports(os.path.join(CSRC, f), prefix):
            if name in skip or one_liner:      # (internal / a body that cannot throw)
                continue
            assert guarded, "%s: %s is not a function-try-block" % (f, name)
            macro = "PCDH_ABI_CATCH" if prefix == "pcdh_" else "PCD_ABI_CATCH"
            assert "} %s(%s)" % (macro, name) in text, (f, name)
            seen += 1
    # 62 engine exports with a status + the host library's
    assert seen >= 62 + 20, seen
    # and the binding knows the new status
    hdr = open(os.path.join(ROOT, "include", "pcd_engine.h")).read()
    assert "PCD_ERR_INTERNAL = 7" in hdr


_HOST_CHILD = r"""
import ctypes, os, resource, sys
import numpy as np
sys.path.insert(0, %(root)r)
os.environ["OMP_NUM_THREADS"] = "4"
from fenapack_amd import _host as H
L = H.library()
import scipy.sparse as sp
# a product whose scratch (one b_cols-sized map per thread) and whose grouped
# pairs do not fit the address space left
n = 200_000
A = sp.diags([np.ones(n - 1), np.ones(n), np.ones(n - 1)], [-1, 0, 1], format="csr")
rows = np.repeat(np.arange(n, dtype=np.int64), 8)
cols = (rows * 7 + 3) %% n
out = ctypes.c_void_p()
# leave ~64 MB above what is mapped now
with open("/proc/self/statm") as f:
    vm = int(f.read().split()[0]) * os.sysconf("SC_PAGE_SIZE")
resource.setrlimit(resource.RLIMIT_AS, (vm + (64 << 20), vm + (64 << 20)))
codes = []
big = 1 << 36                       # 64 G pairs asked for: the vectors cannot be had
i64 = ctypes.POINTER(ctypes.c_int64)
rc = L.pcdh_group_pairs(ctypes.c_int64(rows.size), rows.ctypes.data_as(i64),
                        cols.ctypes.data_as(i64), ctypes.c_int64(big),
                        ctypes.c_int64(0), ctypes.c_int64(big), ctypes.byref(out))
codes.append((rc, L.pcdh_last_error().decode()))
# per-thread scratch INSIDE an OpenMP region (b_cols = 2^33 columns claimed)
arp = A.indptr.astype(np.int32); ac = A.indices.astype(np.int32)
crp = np.zeros(n + 1, dtype=np.int64)
i32 = ctypes.POINTER(ctypes.c_int32)
rc = L.pcdh_spgemm_count(ctypes.c_int64(0), ctypes.c_int64(n), ctypes.c_int64(1 << 33),
                         arp.ctypes.data_as(i32), ac.ctypes.data_as(i32),
                         arp.ctypes.data_as(i32), ac.ctypes.data_as(i32),
                         crp.ctypes.data_as(i64))
codes.append((rc, L.pcdh_last_error().decode()))
print("CODES", codes)
# the library still works afterwards
g = H.group_pairs(rows[:1000], cols[:1000], n)
print("ALIVE", g.nnz)
"""


def test_host_export_reports_a_failed_allocation_as_a_status():
    r = subprocess.run([sys.executable, "-c", _HOST_CHILD % {"root": ROOT}],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    m = re.search(r"CODES (.*)", r.stdout)
    codes = eval(m.group(1))
    for rc, msg in codes:
        assert rc == 2, codes                      # PCDH_ERR_NOMEM
        assert "memory" in msg, codes
    assert re.search(r"ALIVE \d+", r.stdout)


def test_host_library_under_asan_and_ubsan():
    """csrc/Makefile `asan`: the product's own host library (not only the
    oracle) under AddressSanitizer + UBSan, driven by its native test file."""
    lib = os.path.join(ROOT, "fenapack_amd", "lib", "libpcd_host_asan.so")
    subprocess.check_call(["make", "-s", "-C", CSRC, "asan"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"],
                                   text=True).strip()
    ubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"],
                                    text=True).strip()
    if not (os.path.isabs(asan) and os.path.exists(asan)):
        pytest.fail("libasan.so not found next to gcc")
    env = dict(os.environ)
    env.update(FENAPACK_AMD_HOST_LIBRARY=lib,
               LD_PRELOAD=asan + (":" + ubsan if os.path.isabs(ubsan) else ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="4", FENAPACK_AMD_GUARD="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p",
                        "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_host_native.py")],
                       env=env, capture_output=True, text=True, timeout=900,
                       cwd=ROOT)
    tail = (r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr, tail
    assert "runtime error" not in r.stderr, tail


_GPU_CHILD = r"""
import ctypes, os, resource, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, %(root)r)
import torch
from fenapack_amd import _cabi as c
lib = c.hip_library()
e = c.Engine(lib, "BRM1", 0)
n = 1_500_000
d = np.full(n, 4.0); o = np.full(n - 1, -1.0)
A = sp.diags([o, d, o], [-1, 0, 1], format="csr")
e.set_csr(c.MAT_MP, A)                 # the runtime's own mappings exist now
with open("/proc/self/statm") as f:
    vm = int(f.read().split()[0]) * os.sysconf("SC_PAGE_SIZE")
resource.setrlimit(resource.RLIMIT_AS, (vm + (8 << 20), vm + (8 << 20)))
rp = A.indptr.astype(np.int32); ci = A.indices.astype(np.int32); va = A.data
i32 = ctypes.POINTER(ctypes.c_int32); f64 = ctypes.POINTER(ctypes.c_double)
rcs = []
for which in (c.MAT_AP, c.MAT_KP, c.MAT_MP):
    # (the raw export, not the wrapper that raises: the STATUS is the subject)
    rc = lib.lib.pcd_set_csr(e._h, which, ctypes.c_int64(n), ctypes.c_int64(n),
                             rp.ctypes.data_as(i32), ci.ctypes.data_as(i32),
                             va.ctypes.data_as(f64))
    rcs.append((rc, lib.last_error().decode()))
print("CODES", rcs)
rc = lib.lib.pcd_destroy(e._h); e._h = ctypes.c_void_p()
print("DESTROYED", rc)
"""


@pytest.mark.gpu
def test_engine_export_reports_a_failed_allocation_as_a_status():
    """pcd_set_csr when the host side cannot allocate (address-space limit set
    after the HIP runtime is up): a status code and a message - PCD_ERR_NOMEM
    from the catch at the ABI, or PCD_ERR_HIP when the runtime's own
    allocation is what failed -, never a signal; pcd_destroy still works."""
    r = subprocess.run([sys.executable, "-c", _GPU_CHILD % {"root": ROOT}],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    codes = eval(re.search(r"CODES (.*)", r.stdout).group(1))
    assert any(rc != 0 for rc, _ in codes), codes
    for rc, msg in codes:
        assert rc in (0, 2, 3, 7), codes
        if rc:
            assert msg, codes
    assert "DESTROYED 0" in r.stdout
