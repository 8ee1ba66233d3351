"""Import the package from the repository without installing it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as _entry  # noqa: E402

pkg = _entry.load_package()
TinyMPC = pkg.TinyMPC
problems = pkg.problems
