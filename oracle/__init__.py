"""TEST INFRASTRUCTURE: CPU oracle of the anchor-search path (see oracle.py)."""
