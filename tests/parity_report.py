"""Collects the measured parity numbers of a test session (mismatch counts, maximum errors) so that the session ends
with ONE line ``PARITY_REPORT {...}`` in the log: a run that passes shows its counts, not dots (tests/conftest.py
prints it from ``pytest_terminal_summary``)."""
RESULTS = {}


def record(name, **values):
    RESULTS[name] = values
