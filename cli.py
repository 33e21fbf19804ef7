#!/usr/bin/env python3
"""python cli.py ... -- the reference's command line (/root/reference/cli.py) on the MI355X package, offline."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from subword_tokenizers_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    raise SystemExit(main())
