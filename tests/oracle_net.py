"""Kept for the tests' import path: the oracle-backed 'net' lives in oracle/net.py (test infrastructure)."""
from oracle.net import OracleNet  # noqa: F401
