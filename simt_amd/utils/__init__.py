"""Drop-in mirror of the reference's `utils` package (utils/loss.py)."""
