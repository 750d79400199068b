"""Second half of tools/sanitize_cpu.sh: tests/test_abi.py against the asan_host build of the library (the process
must have clang's ASan runtime preloaded).  The variant is bound before pytest imports anything."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest  # noqa: E402
from dfu3d_amd import _lib  # noqa: E402

_lib._LIB = _lib.load_variant("asan_host")
assert "asan_host" in _lib._LIB._name
sys.exit(pytest.main(["tests/test_abi.py", "-q", "-x", "-p", "no:cacheprovider"]))
