"""CPU oracle for the beacon hot path -- TEST INFRASTRUCTURE ONLY (see oracle.py)."""
