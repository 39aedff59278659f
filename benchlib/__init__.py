"""Bench / test tooling that is not part of the product: the seeded synthetic read generator."""
