"""Mirrors of the reference's `util/` pieces that sit next to the scoring path."""
