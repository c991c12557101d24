"""CPU oracle for the NFAI Llama-3 decode path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker / reported baseline.  ``nfai_amd`` never imports it.

PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures for this path and
cannot be built here (C#/.NET 9 + Vulkan); see ``nfai_oracle.c`` header and DESIGN.md.

* ``oracle.c_oracle``  — ctypes binding of ``nfai_oracle.c`` (fp32, reference summation order).
* ``oracle.np_oracle`` — independent fp64 NumPy evaluation used to cross-check the C file.
"""
from .c_oracle import *  # noqa: F401,F403
from . import np_oracle  # noqa: F401
