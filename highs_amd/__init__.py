"""highs_amd — MI355X-native PDLP hot path for HiGHS (solver="pdlp").

The product is the C-ABI shared library built from highs_amd/csrc
(include/pdlp_mi355x.h); this package is the thin host-side mirror used by
tests and bench: ctypes structs (abi), the HighsLp container / MPS reader (lp)
and the solver entry points (solver)."""
from . import abi, lp  # noqa: F401
