"""Drop-in mirrors of the reference's ``src/features`` extractors (same names and signatures), and
``extract_all``: every feature set of a batch from one pack and one engine call."""
from ._frame import extract_all  # noqa: F401
