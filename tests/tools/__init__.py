"""Debug / analysis aids that use the oracle (test infrastructure: they live under tests/ because only tests may import oracle/)."""
