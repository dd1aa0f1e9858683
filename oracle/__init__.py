"""CPU oracle package (TEST INFRASTRUCTURE ONLY -- see oracle/oracle.c header)."""
