"""Drop-in mirrors of the reference's ``src/features`` extractors (same names and signatures)."""
