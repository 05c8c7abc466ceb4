"""Shared helpers for the tests: library loading and ctypes mirrors of include/pimgpu.h."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE = None


def load_oracle():
    """TEST INFRASTRUCTURE ONLY: the CPU oracle (oracle/liborc.so)."""
    global _ORACLE
    if _ORACLE is None:
        path = os.path.join(ROOT, "oracle", "liborc.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _ORACLE = ctypes.CDLL(path)
    return _ORACLE
