#!/usr/bin/env python3
"""Command line of pyro_amd/kernel_manifest.py (code-object provenance: build / diff / check / bless)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyro_amd import kernel_manifest  # noqa: E402

if __name__ == "__main__":
    kernel_manifest.main()
