"""oracle — TEST INFRASTRUCTURE ONLY (see oracle/ctr_oracle.c).  Python bindings of the two CPU checkers.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and only
as the checker / reported CPU baseline.  Nothing under cutrace_amd/ imports it.
"""
from .bindings import (oracle_lib, oracle_render, ref_cudaminmax_lib, ref_cudaminmax_render, ref_fmad_lib, ref_fmad_render,  # noqa: F401
                       ref_lib, ref_render)
