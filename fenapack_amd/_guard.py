"""Host-memory guards for everything that travels to a GPU box.

Two boxes were lost in round 3 to host code (a 1.35 TB ``scipy.sparse.random``
and eight thread ranks each building a 10 M-DOF problem), not to a kernel.
The operator producer of this package is host code in front of the engine
(the reference's is DOLFIN: ``fenapack/assembling.py:151-180``), so the guard
lives with it:

* ``host_memory_available()`` - what this process may still allocate
  (control-group aware; page cache is not counted as used);
* ``start_rss_watchdog()`` - a daemon thread that polls the resident set of
  this process and ends it with a non-zero status (``os._exit``: never a
  re-exec, never a signal to another process) before the kernel's OOM killer
  or the box does.  ``bench.py``, ``tests/conftest.py`` and every script under
  ``tools/`` start it first thing.
"""
import os
import sys
import threading
import time

_PAGE = os.sysconf("SC_PAGE_SIZE") if hasattr(os, "sysconf") else 4096

#: exit status of a process the watchdog ended
WATCHDOG_EXIT = 97


def rss_bytes():
    """Resident set of this process now (``/proc/self/statm``)."""
    try:
        with open("/proc/self/statm") as f:
            return int(f.read().split()[1]) * _PAGE
    except (OSError, ValueError, IndexError):
        return 0


def peak_rss_bytes():
    """High-water mark of the resident set (``VmHWM``)."""
    try:
        with open("/proc/self/status") as f:
            for line in f:
                if line.startswith("VmHWM:"):
                    return int(line.split()[1]) * 1024
    except (OSError, ValueError):
        pass
    import resource
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss * 1024


def _cgroup_available():
    """limit - (current - reclaimable file cache) of the control group, or
    None.  ``memory.current`` counts page cache, which the kernel drops under
    pressure: after file-heavy work it would under-state what is left."""
    for limit_path, used_path, stat_path in (
            ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current",
             "/sys/fs/cgroup/memory.stat"),
            ("/sys/fs/cgroup/memory/memory.limit_in_bytes",
             "/sys/fs/cgroup/memory/memory.usage_in_bytes",
             "/sys/fs/cgroup/memory/memory.stat")):
        try:
            v = open(limit_path).read().strip()
            if v == "max" or int(v) >= (1 << 60):
                continue
            used = int(open(used_path).read().strip())
            st = {}
            try:
                for line in open(stat_path):
                    k, _, val = line.partition(" ")
                    st[k] = int(val)
            except (OSError, ValueError):
                st = {}
            if "total_inactive_file" in st:          # (v1, hierarchical)
                cache = st["total_inactive_file"] + st.get("total_active_file", 0)
            else:
                cache = st.get("inactive_file", 0) + st.get("active_file", 0)
            return int(v) - max(0, used - cache)
        except (OSError, ValueError):
            continue
    return None


def host_memory_available():
    """Bytes this process may still allocate: the smaller of the control
    group's head-room (a container usually owns a fraction of the machine
    ``free`` shows) and the kernel's MemAvailable; ``None`` when unknown."""
    cands = []
    cg = _cgroup_available()
    if cg is not None:
        cands.append(cg)
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                cands.append(int(line.split()[1]) * 1024)
                break
    except OSError:
        pass
    return min(cands) if cands else None


def concurrent_builds():
    """How many producer builds share this host's memory: the thread-rank
    tools and ``bench.py``'s launcher export FENAPACK_AMD_CONCURRENT_BUILDS;
    under ``torch.distributed.run`` LOCAL_WORLD_SIZE says the same."""
    n = os.environ.get("FENAPACK_AMD_CONCURRENT_BUILDS")
    if n is None:
        n = os.environ.get("LOCAL_WORLD_SIZE", "1")
    try:
        return max(1, int(n))
    except ValueError:
        return 1


_WATCHDOG = {"thread": None, "limit": None, "peak": 0}


def watchdog_limit_bytes(processes=1):
    """Default ceiling of one process: half of what is available when the
    watchdog starts, shared among ``processes`` of the same job; never above
    FENAPACK_AMD_RSS_LIMIT_GB when that is set."""
    have = host_memory_available()
    limit = None if have is None else 0.5 * have / max(1, processes)
    env = os.environ.get("FENAPACK_AMD_RSS_LIMIT_GB")
    if env:
        cap = float(env) * 1e9
        limit = cap if limit is None else min(limit, cap)
    return limit


def _native_watchdog(limit_bytes, interval):
    """The poller as a native thread of libpcd_guard.so (no GIL: a numpy call
    that allocates while holding it cannot starve the watchdog).  False when
    the library is not built yet."""
    import ctypes
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                        "libpcd_guard.so")
    if not os.path.exists(path):
        return False
    try:
        L = ctypes.CDLL(path)
        L.pcdg_watchdog_start.argtypes = [ctypes.c_int64, ctypes.c_int,
                                          ctypes.c_int]
        L.pcdg_watchdog_peak.restype = ctypes.c_int64
        rc = L.pcdg_watchdog_start(int(limit_bytes),
                                   max(1, int(interval * 1000)), WATCHDOG_EXIT)
    except (OSError, AttributeError):
        return False
    if rc != 0:
        return False
    _WATCHDOG["native"] = L
    return True


def watchdog_peak_bytes():
    """Highest resident set the watchdog has seen (0: not started)."""
    L = _WATCHDOG.get("native")
    if L is not None:
        return int(L.pcdg_watchdog_peak())
    return max(_WATCHDOG["peak"], rss_bytes()) if _WATCHDOG["thread"] else 0


def start_rss_watchdog(limit_bytes=None, processes=None, interval=0.05,
                       what=None):
    """Start (once per process) the thread that ends this process with status
    ``WATCHDOG_EXIT`` when its resident set exceeds ``limit_bytes`` (default:
    ``watchdog_limit_bytes``).  Returns the limit in force (None: unknown
    host, no watchdog).  The thread is native (``pcdg_watchdog_start``) when
    libpcd_guard.so is built, a Python daemon thread otherwise."""
    if _WATCHDOG["thread"] is not None:
        return _WATCHDOG["limit"]
    if os.environ.get("FENAPACK_AMD_NO_WATCHDOG") == "1":
        return None
    if limit_bytes is None:
        limit_bytes = watchdog_limit_bytes(
            concurrent_builds() if processes is None else processes)
    if limit_bytes is None:
        return None
    if _native_watchdog(limit_bytes, interval):
        _WATCHDOG["thread"], _WATCHDOG["limit"] = "native", limit_bytes
        return limit_bytes
    name = what or os.path.basename(sys.argv[0] or "python")

    def poll():
        while True:
            r = rss_bytes()
            if r > _WATCHDOG["peak"]:
                _WATCHDOG["peak"] = r
            if r > limit_bytes:
                try:
                    sys.stderr.write(
                        "\n%s: RSS watchdog: resident set %.1f GB exceeds the "
                        "limit of %.1f GB (half of the host memory available "
                        "at start, per process; FENAPACK_AMD_RSS_LIMIT_GB) - "
                        "ending this process with status %d before the host "
                        "runs out of memory\n"
                        % (name, r / 1e9, limit_bytes / 1e9, WATCHDOG_EXIT))
                    sys.stderr.flush()
                finally:
                    os._exit(WATCHDOG_EXIT)
            time.sleep(interval)

    t = threading.Thread(target=poll, name="rss-watchdog", daemon=True)
    t.start()
    _WATCHDOG["thread"], _WATCHDOG["limit"] = t, limit_bytes
    return limit_bytes


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main_script_is_ours():
    """Is the running program one of this repository's scripts (bench.py,
    tools/, demo/, tests/ run directly) rather than a host application?"""
    main = sys.modules.get("__main__")
    path = getattr(main, "__file__", None) or (sys.argv[0] if sys.argv else "")
    if not path or not os.path.isfile(path):
        return False
    try:
        path = os.path.realpath(path)
    except OSError:
        return False
    return path.startswith(os.path.realpath(ROOT) + os.sep)


def autostart():
    """Called at ``import fenapack_amd``: start the watchdog for this
    repository's own scripts, or anywhere with FENAPACK_AMD_WATCHDOG=1."""
    if os.environ.get("FENAPACK_AMD_WATCHDOG") == "1" or main_script_is_ours():
        return start_rss_watchdog()
    return None
