"""Synthetic vectorised environments (Isaac-Gym stand-ins)."""
