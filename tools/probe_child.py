#!/usr/bin/env python3
"""Does a GPU-initialised process start child processes on this box? (decides whether a -m gpu test may launch
tools/dist_rehearsal.py; the child is a fresh interpreter, nothing is exec'ed in THIS process)"""
import subprocess
import sys

import torch

torch.zeros(4, device="cuda").sum().item()
p = subprocess.run([sys.executable, "-c", "print('child ok')"], capture_output=True, text=True, timeout=120)
print("rc", p.returncode, "out", p.stdout.strip(), "err", p.stderr.strip()[-300:])
