"""pql_amd -- MI355X-native Parallel-Q-Learning learner.

Mirrors the module layout of the reference package `pql` (replay/, models/, algo/, utils/, cfg/) so the
reference's entry points and class-name plugin lookup keep working; all hot-path math runs in
libpqlk.so (hand-written HIP for gfx950, see include/pqlk.h).
"""
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent  # reference: pql/__init__.py:3

__version__ = "0.1.0"
