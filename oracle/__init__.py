"""CPU oracle of the cqs hot path — TEST INFRASTRUCTURE ONLY (see oracle/cqs_oracle.h)."""
