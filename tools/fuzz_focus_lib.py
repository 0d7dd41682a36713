"""tools/fuzz_focus.py on the library LFI_AB_LIB names (measurement builds of the focus-map kernels must give the oracle's maps too)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import _ablib  # noqa: F401
exec(open("tools/fuzz_focus.py").read())
