"""bench.py's machinery (the entry point and the CPU-baseline legs stay in /bench.py): see bench.py's docstring."""
