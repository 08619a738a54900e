"""Drop-in for the third-party `steerable` package the reference imports (src/train/pyramid.py:7-8)."""
