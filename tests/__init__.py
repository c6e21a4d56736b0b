"""Parity tests (``-m gpu`` through the C ABI) and CPU-side oracle / host-logic tests."""
