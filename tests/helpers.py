"""Shared builders for the parity tests (the implementation lives with the oracle: oracle/mirror.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.mirror import oracle_from_scene, rel_err, sync_oracle_state  # noqa: E402,F401
