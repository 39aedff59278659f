"""Drop-in module name of the reference (`from vfind import find_variants`, reference
tests/test_vfind.py:1): re-exports the MI355X-native implementation."""
from vfind_amd import find_variants  # noqa: F401

__all__ = ["find_variants"]
