"""Shares the fixtures of tests/conftest.py (knobs, the gpu marker) with the experiment tests."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import *  # noqa: F401,F403,E402
from tests.conftest import knobs, pytest_configure  # noqa: F401,E402
