"""hyrise_amd -- MI355X-native execution hot path for Hyrise (TableScan / JoinHash / AggregateHash).

The product is the C-ABI library libhyrise_amd.so (include/hyrise_amd.h); this package is the thin Python plumbing
around it used by tests, bench.py and the multi-GPU launcher.  Importing `hyrise_amd.abi.load_library()` fails loudly
when the HIP library has not been built -- there is no CPU fallback.
"""
from . import abi  # noqa: F401

__all__ = ["abi", "storage", "operators"]
