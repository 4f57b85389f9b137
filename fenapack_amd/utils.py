"""Small helpers mirrored from ``fenapack/utils.py``.

``get_default_factor_solver_type`` / ``pc_set_factor_solver_type``
(``utils.py:7-34``) have no counterpart: sparse direct factorisations are
PETSc/MUMPS host code and are out of scope.
"""

import functools


def allow_only_one_call(method):
    """Let an instance method run once per instance; later calls raise
    ``RuntimeError`` (behaviour of ``fenapack/utils.py:37-60``, which guards
    ``PCDKSP.init_pcd`` at ``field_split.py:60``)."""
    flag = "_called_once__" + method.__name__

    @functools.wraps(method)
    def guarded(self, *args, **kwargs):
        if self.__dict__.get(flag, False):
            raise RuntimeError("Multiple calls to %s.%s not allowed"
                               % (type(self).__name__, method.__name__))
        self.__dict__[flag] = True
        return method(self, *args, **kwargs)

    return guarded
