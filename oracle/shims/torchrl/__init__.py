"""Test-only stand-in for `torchrl` (see oracle/shims/README.md)."""
