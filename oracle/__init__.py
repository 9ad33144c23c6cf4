"""CPU restatements of the reference's hot path.  TEST INFRASTRUCTURE ONLY (see oracle/oracle.c)."""
