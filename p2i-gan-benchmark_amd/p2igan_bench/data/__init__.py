"""Data layer."""
