"""Short fixed-seed runs of the emulator fuzzers / stress tools (tools/emu_*.py) so that they stay runnable and their
properties stay checked by the CPU suite; the long runs are recorded in DESIGN.md section 7."""
import os
import subprocess
import sys

import pytest

import zmi_ctypes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool, seed", [("emu_fuzz_inflate.py", 3), ("emu_fuzz_stream.py", 4), ("emu_fuzz_deflate_calls.py", 5),
                                        ("emu_fuzz_inflate_calls.py", 6), ("emu_stress_mixed.py", 7), ("emu_stress.py", 8)])
def test_fuzz_tool_short_run(tool, seed):
    zmi_ctypes.load_emu()          # the tools load the emulator library without rebuilding it
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(seed), "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
