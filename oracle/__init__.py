"""CPU oracle for the NUFFT hot path -- test infrastructure only (see nufft_oracle.c)."""
