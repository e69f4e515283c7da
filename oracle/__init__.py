"""Test infrastructure only: CPU oracle for the aggforce hot path (see aggforce_oracle.py)."""
