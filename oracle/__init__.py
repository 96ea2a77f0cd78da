"""CPU oracle of the MI355X engine: test infrastructure only (see oracle/oracle.py)."""
