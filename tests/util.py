"""Shared helpers for the tests: library loading and ctypes mirrors of include/pimgpu.h."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE = {}


def load_oracle(portable=False):
    """TEST INFRASTRUCTURE ONLY: the CPU oracle.

    portable=False: oracle/liborc.so   (glibc libm; bit-exact to the reference's headers)
    portable=True:  oracle/liborc_p.so (portable libm of oracle/orc_libm.h; the variant the
                    HIP path is compared with bit for bit)"""
    name = "liborc_p.so" if portable else "liborc.so"
    if name not in _ORACLE:
        path = os.path.join(ROOT, "oracle", name)
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _ORACLE[name] = ctypes.CDLL(path)
    return _ORACLE[name]
