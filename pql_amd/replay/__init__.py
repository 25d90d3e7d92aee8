"""Replay ring and n-step assembler on HBM."""
