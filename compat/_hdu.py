"""resolves the real package (directory name contains a hyphen, so it is imported through importlib)"""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)


def mod(name):
    return importlib.import_module("h-denseunet_amd." + name)
