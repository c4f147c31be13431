"""DESI driver (ingestion / egress) -- mirror of py/rvspecfit/desi/."""
