"""Host micro-timings on the GPU box: native SpMV/SpMM by thread count, dense
inverse and skinny GEMM under BLAS thread limits."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenapack_amd import _host
from threadpoolctl import threadpool_info, threadpool_limits


def best(f, n=5):
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        t.append(time.perf_counter() - t0)
    return min(t) * 1e3, sorted(t)[len(t) // 2] * 1e3


print([(d["internal_api"], d["num_threads"], d.get("threading_layer"))
       for d in threadpool_info()])
# (a banded stand-in for the scalar velocity operator of cavity level 6: 12
# entries per row at fixed offsets - built directly, O(nnz) memory)
n = 411000
rng = np.random.default_rng(0)
offs = np.array([-1300, -1282, -641, -640, -2, -1, 0, 1, 2, 640, 641, 1282])
cols = (np.arange(n)[:, None] + offs[None, :]) % n
cols.sort(axis=1)
A = sp.csr_matrix((rng.standard_normal(cols.size), cols.ravel().astype(np.int32),
                   np.arange(n + 1, dtype=np.int32) * offs.size), shape=(n, n))
x2 = rng.standard_normal(2 * n)
sc = rng.standard_normal(2 * n)
L = _host.library()
for T in (1, 4, 8, 16, 32, 64):
    L.pcdh_set_threads(T)
    op = _host.SpMV(A, sc, nvec=2)
    print("spmm nvec=2 threads %3d: min %.2f ms median %.2f ms" % ((T,) + best(lambda: op(x2))))
L.pcdh_set_threads(0)
print("scipy csr_matvecs: %.2f / %.2f ms" % best(lambda: A @ x2.reshape(-1, 2)))
D = rng.standard_normal((441, 441)) + 30 * np.eye(441)
D8 = rng.standard_normal((882, 882)) + 30 * np.eye(882)
G1 = rng.standard_normal((204800, 18))
G2 = rng.standard_normal((18, 36))
for lim in (None, 1, 4, 8, 16, 32):
    ctx = threadpool_limits(limits=lim) if lim else threadpool_limits(limits=None)
    with ctx:
        a = best(lambda: np.linalg.inv(D))
        b = best(lambda: np.linalg.inv(D8))
        g = best(lambda: G1 @ G2)
    print("blas limit %s: inv441 %.2f/%.2f ms  inv882 %.2f/%.2f ms  gemm %.2f/%.2f ms"
          % ((lim,) + a + b + g))
t0 = time.perf_counter()
with threadpool_limits(limits=8):
    pass
print("threadpool_limits enter/exit %.3f ms" % ((time.perf_counter() - t0) * 1e3))
