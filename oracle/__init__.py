"""CPU oracle for the VALL-E inference hot path — test infrastructure only (see valle_oracle.py)."""
