"""Parity oracle -- TEST INFRASTRUCTURE only (see oracle/ljmd_oracle.c)."""
