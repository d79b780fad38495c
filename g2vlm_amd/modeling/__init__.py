"""Mirror of the reference's `modeling` package surface (reference modeling/__init__.py)."""
from . import g2vlm  # noqa: F401
