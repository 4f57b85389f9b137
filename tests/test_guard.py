"""CPU suite: the host-memory guards (fenapack_amd/_guard.py,
include/pcd_guard.h) - the answer to the two GPU boxes lost in round 3 to host
allocations of this repository's own scripts."""
import ctypes
import glob
import os
import re
import subprocess
import sys

from fenapack_amd import _guard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_guard_library_exports_every_symbol_of_the_header():
    text = open(os.path.join(ROOT, "include", "pcd_guard.h")).read()
    names = sorted(set(re.findall(r"\b(pcdg_[a-z_0-9]+)\s*\(", text)))
    assert names == ["pcdg_watchdog_limit", "pcdg_watchdog_peak",
                     "pcdg_watchdog_start"]
    lib = ctypes.CDLL(os.path.join(ROOT, "fenapack_amd", "lib",
                                   "libpcd_guard.so"))
    for n in names:
        assert hasattr(lib, n), n
    # no OpenMP runtime comes with it (it is loaded first thing in a process)
    out = subprocess.run(["ldd", lib._name], capture_output=True, text=True)
    assert "gomp" not in out.stdout and "omp" not in out.stdout, out.stdout


def test_watchdog_ends_a_process_that_outgrows_its_limit():
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from fenapack_amd import _guard as g\n"
        "assert g.start_rss_watchdog(limit_bytes=1.5e9) == 1.5e9\n"
        "assert g._WATCHDOG['thread'] == 'native'\n"
        "keep = []\n"
        "for i in range(40):\n"
        "    keep.append(np.ones(1 << 25))\n"       # 256 MiB each, touched
        "print('survived', g.rss_bytes())\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == _guard.WATCHDOG_EXIT, (out.returncode, out.stderr)
    assert "RSS watchdog" in out.stderr and "survived" not in out.stdout


def test_available_memory_and_limits_are_sane():
    have = _guard.host_memory_available()
    assert have is None or 1e8 < have < 1e14
    if have is not None:
        assert abs(_guard.watchdog_limit_bytes(4) - 0.125 * have) < 0.02 * have
    assert _guard.rss_bytes() > 1e6
    assert _guard.peak_rss_bytes() >= _guard.rss_bytes() * 0.5


def test_every_script_that_travels_runs_under_the_watchdog():
    """bench.py, the tools and the demos import fenapack_amd (whose import
    starts the watchdog for this repository's own scripts) or the guard
    itself; the suite starts it in conftest.pytest_sessionstart."""
    scripts = glob.glob(os.path.join(ROOT, "tools", "*.py")) \
        + glob.glob(os.path.join(ROOT, "demo", "*.py")) \
        + [os.path.join(ROOT, "bench.py")]
    assert len(scripts) > 20
    for path in scripts:
        text = open(path).read()
        assert re.search(r"^\s*(from|import) fenapack_amd", text, re.M), path
    assert "start_rss_watchdog" in open(
        os.path.join(ROOT, "tests", "conftest.py")).read()
    # the launcher tells every rank how many builds share the host
    assert "FENAPACK_AMD_CONCURRENT_BUILDS" in open(
        os.path.join(ROOT, "bench.py")).read()
    assert not glob.glob(os.path.join(ROOT, "tools", "r0*_*.sh")), \
        "one-off launch scripts belong in gpurun_out/, not in tools/"


def test_autostart_only_for_this_repository_s_scripts():
    """`import fenapack_amd` in a host application leaves the process alone."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import fenapack_amd\n"
            "from fenapack_amd import _guard\n"
            "print(_guard._WATCHDOG['thread'])\n" % ROOT)
    env = {k: v for k, v in os.environ.items()
           if k != "FENAPACK_AMD_WATCHDOG"}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True,
                         text=True, env=env, timeout=120, cwd="/tmp")
    assert out.stdout.strip() == "None", (out.stdout, out.stderr[-500:])
    out = subprocess.run([sys.executable, "-c", code], capture_output=True,
                         text=True, env=dict(env, FENAPACK_AMD_WATCHDOG="1"),
                         timeout=120, cwd="/tmp")
    assert out.stdout.strip() == "native", (out.stdout, out.stderr[-500:])
