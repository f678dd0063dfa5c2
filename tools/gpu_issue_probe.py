#!/usr/bin/env python3
"""Cycles per wave64 instruction on the MI355X: builds tools/issue_probe.hip with hipcc if the binary
is missing and runs it; the table goes to stdout (tools/gpu_issue_probe.py > gpurun_out/<tag>/issue_probe.txt, then copied
to profiles/).  Standalone HIP program: no torch, no product library."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
exe = os.path.join(HERE, "issue_probe")
src = os.path.join(HERE, "issue_probe.hip")
if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O2", "-o", exe, src], check=True)
sys.exit(subprocess.run([exe] + sys.argv[1:]).returncode)
