"""CPU oracle (test infrastructure only; see the header of tamcmc_oracle.c)."""
